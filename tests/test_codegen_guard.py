"""CPU tier: regression guard on the code hipcc generates for the hot kernels (no GPU needed).

The shipped kernel's speed rests on two compiler-facing tricks in hades_fast.hpp (`pin`, `limb_fence`): with
them every limb product is ONE `v_mad_i64_i32`; without them LLVM widens limbs to 64 bits (two multiply-adds
and two moves per product) or re-associates column sums (an extra 64-bit add per column).  A ROCm bump could
silently undo either.  This test compiles the device code to assembly for gfx950 and asserts, per kernel:
  k_perm_fast      v_mad_i64_i32 within +-2 % of 2562 (round 4: constant products as linear maps, 97 multiply-adds
                   instead of 153), no v_mad_u64_u32 beyond the staging/finalize few, 0 scratch, <= 112 VGPRs (=> 4
                   waves/SIMD by registers; the LDS slab admits 3), the linear map's multipliers fetched a column at a
                   time (no dword-by-dword scalar loads)
  every hot kernel 0 scratch (sponge, Merkle, trace, cooperative, per-lane level) -- round 1's sponge spilled
  k_perm_witness   one multiply-add per limb product in its rolled loops too (no hoisted tables, no widened operands)
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hades252_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "hades252.s"
    r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S",
                        "-o", str(out), "hades252.hip", "-Rpass-analysis=kernel-resource-usage"],
                       cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    text = open(out).read()
    # kernel bodies: from "<name>:" to the end of the function (a kernel may hold several s_endpgm: early exits)
    bodies = {}
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\n\.Lfunc_end", text, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    # resource remarks
    res, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return bodies, res


def find(d, needle):
    hits = [k for k in d if needle in k]
    assert len(hits) >= 1, "kernel %s not found in %s" % (needle, sorted(d)[:50])
    return hits


def test_perm_fast_instruction_mix(device_asm):
    bodies, res = device_asm
    (name,) = find(bodies, "k_perm_fast")
    body = bodies[name]
    mads = len(re.findall(r"\bv_mad_i64_i32\b", body))
    umads = len(re.findall(r"\bv_mad_u64_u32\b", body))
    assert abs(mads - 2562) <= 0.02 * 2562, "one multiply-add per limb product no longer holds: %d v_mad_i64_i32" % mads
    assert umads <= 300, "%d v_mad_u64_u32: limbs are being widened to 64 bits?" % umads
    total = len([l for l in body.splitlines() if re.match(r"\s+[vs]_", l)])
    assert total < 5600, "kernel grew to %d instructions (I-cache: 64 KB ~ 8 k instructions of 8 B)" % total
    r = res[name]
    assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0
    assert r["VGPRs"] <= 112, r
    # mont_lin's 81 multipliers per partial round arrive as one s_load_dwordx8 + one s_load_dword per column, a column
    # ahead; left to itself the scheduler fetched them dword by dword, each behind its own wait
    singles = len(re.findall(r"\bs_load_dword\b", body))
    assert singles <= 24, "%d single-dword scalar loads: the column staging of mont_lin got lost" % singles


@pytest.mark.parametrize("needle", ["k_perm_fast", "k_sponge", "k_merkle_level_fast", "k_merkle_coop", "k_perm_coop",
                                    "k_perm_trace_fast", "k_perm_trace_scaled", "k_perm_witness", "k_fr_op", "k_perm_lanes", "k_merkle_lanes",
                                    "k_merkle_verify", "k_merkle_update", "k_wire", "k_perm_rows", "k_merkle_rows"])
def test_hot_kernels_have_no_scratch(device_asm, needle):
    _, res = device_asm
    for name in find(res, needle):
        assert res[name]["ScratchSize"] == 0, (name, res[name])
        assert res[name]["VGPRs Spill"] == 0, (name, res[name])      # (SGPR spills go to VGPR lanes, not memory)


def test_scaled_trace_kernel_is_the_throughput_round_plus_a_cheap_exit(device_asm):
    """k_perm_trace_scaled = the round body of k_perm_fast (one per round kind) + five exits per round: its multiply-add count
    is the throughput kernel's minus the final un-scaling maps (5 x 97), no block barrier inside the round loop (the slab is
    wave-private: two s_barrier, both in the initial load), four waves per SIMD by registers."""
    bodies, res = device_asm
    (name,) = find(bodies, "k_perm_trace_scaled")
    (fast,) = find(bodies, "k_perm_fast")
    body = bodies[name]
    mads = len(re.findall(r"\bv_mad_i64_i32\b", body))
    fast_mads = len(re.findall(r"\bv_mad_i64_i32\b", bodies[fast]))
    assert abs(mads - (fast_mads - 5 * 97)) <= 0.02 * fast_mads, (mads, fast_mads)
    assert len(re.findall(r"\bs_barrier\b", body)) == 2
    r = res[name]
    assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0 and r["VGPRs"] <= 128, r


def test_wire_kernels_keep_their_occupancy(device_asm):
    """k_wire is straight-line code: left alone hipcc schedules it for ILP with ~200 VGPRs (2 waves/SIMD, 4.0 TB/s instead
    of 5.9).  The launch bounds hold it at >= 8 (to_bytes) / >= 6 (from_bytes) waves per SIMD."""
    _, res = device_asm
    for name in find(res, "k_wire"):
        want = 6 if "ILi1E" in name else 8
        assert res[name]["Occupancy"] >= want, (name, res[name])


def test_witness_kernel_products_stay_single_multiply_adds(device_asm):
    """k_perm_witness (true-form schedule) is rolled loops around stores that branch on `live`.  Two things went wrong while
    it was written and would again silently: (i) a loop-invariant linear-map table is hoisted out of the loop as 81 64-bit
    values (SGPR spills, every product widened), (ii) a limb whose sign extension was computed in an earlier basic block is
    multiplied as a 64-bit value (v_mul_lo_u32 pairs around a v_mad_u64_u32).  One multiply-add per limb product:
    6 linear maps (97: the input's, rolled, + five in two grouped walks over the table) + S-box (117 + 117 + 153) + r1 row (35) + rows (265) + 10
    finalize32 (9) = 1359."""
    bodies, res = device_asm
    (name,) = find(bodies, "k_perm_witness")
    body = bodies[name]
    mads = len(re.findall(r"\bv_mad_[iu]64_[iu]32\b", body))
    widened = len(re.findall(r"\bv_mul_lo_u32\b", body))
    assert abs(mads - 1359) <= 0.02 * 1359, mads
    assert widened <= 16, "%d v_mul_lo_u32 (store addresses account for ~7): limb products are being widened to 64 x 32 bits" % widened
    assert "flat_load" not in body and "scratch_" not in body
    r = res[name]
    # (three waves per SIMD: enough, measured; a couple of SGPRs parked in VGPR lanes are harmless -- the hoisted table cost 60)
    assert r["SGPRs Spill"] <= 8 and r["VGPRs"] <= 152 and r["ScratchSize"] == 0, r         # 135 today; 168 is where 3 waves end


def test_perm_lanes_instruction_mix(device_asm):
    """The lane-split kernel's latency IS its instruction count (one wave issues one instruction per ~4.4 cycles): a
    product is 28 multiply-adds (9 + 9 + 9 + the column-16 one) and ~45 DPP moves; a partial round runs 3 products, a
    full round 6.  A toolchain change that splits the 64-bit multiply-adds or doubles the DPP traffic fails here."""
    bodies, res = device_asm
    names = find(bodies, "k_perm_lanes")
    assert len(names) == 2                      # the helped form (three states + a helper wave) and the plain one
    for name in names:
        body = bodies[name]
        umads = len(re.findall(r"\bv_mad_u64_u32\b", body))
        dpp = len(re.findall(r"\bv_mov_b32_dpp\b", body))
        nops = len(re.findall(r"\bs_nop\b", body))
        helped = "s_barrier" in body
        # products in the round bodies: full (twice in the code: leading and trailing loops) + partial, x 28 multiply-adds;
        # + the linear-layer rows (6 each), the helper's rounds and the per-lane final product
        assert 9 * 28 <= umads <= 22 * 28 + 80, (name, umads)
        assert dpp <= 22 * 45 + 60, (name, dpp)
        assert nops <= (330 if helped else 260), "%s: %d s_nop: the hand-made interleaving got lost?" % (name, nops)
        assert res[name]["VGPRs"] <= 96, res[name]
