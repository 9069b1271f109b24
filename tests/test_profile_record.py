"""CPU tier: the committed counter record bench.py replays (profiles/hbm_traffic.json) is a MEASUREMENT, keyed to the kernel
it was taken on -- never a hand-edited key."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from hades252_amd import build  # noqa: E402

REC = os.path.join(ROOT, "profiles", "hbm_traffic.json")


def _is_git_checkout():
    try:
        return subprocess.run(["git", "rev-parse", "--git-dir"], cwd=ROOT, capture_output=True).returncode == 0
    except OSError:                      # no git binary at all
        return False


def test_record_carries_no_hand_made_key():
    rec = json.load(open(REC))
    assert "hash_note" not in rec, "a re-keyed record is not a measurement: run tools/profile_round.sh on the final sources"
    assert rec["kernel"] == "k_perm_fast" and rec["perms_per_launch"] == 1 << 26
    assert len(rec["measured_at_commit"]) == 40
    # counters, not literals: traffic within a percent of the algorithmic bytes, instruction count in the kernel's range
    assert 0.99 < rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"] < 1.02
    assert 70000 < rec["valu_instructions_per_wave"] < 100000


def test_record_key_is_the_current_kernel():
    """bench.py replays `traffic` and `valu_issue` only under this equality; the test makes a kernel edit without a new
    measurement visible in the CPU tier instead of silently dropping the fields from the bench line."""
    rec = json.load(open(REC))
    assert rec["kernel_source_hash"] == build.perm_fast_hash(), \
        "k_perm_fast's sources changed after the profile was taken: measure again (tools/profile_round.sh + tools/stamp_profile.py)"


@pytest.mark.skipif(not _is_git_checkout(), reason="needs the git history (the GPU box has none)")
def test_measured_commit_really_holds_the_measured_kernel():
    import stamp_profile
    rec = json.load(open(REC))
    commit = rec["measured_at_commit"]
    assert subprocess.run(["git", "cat-file", "-e", commit + "^{commit}"], cwd=ROOT).returncode == 0, "unknown commit"
    assert stamp_profile.perm_fast_hash_at(commit) == rec["kernel_source_hash"]
    sec = rec.get("secondary_kernels")
    # (a record keyed to other sources than today's -- e.g. taken under an older recipe of the key -- is simply not
    #  replayed by bench.py: `traffic` null.  Only a record that CLAIMS today's sources is held to its commit.)
    if sec and sec["device_source_hash"] == build.device_source_hash():
        assert stamp_profile.device_source_hash_at(sec["measured_at_commit"]) == sec["device_source_hash"]
        for key in ("wire_to_bytes", "wire_from_bytes", "witness", "trace", "trace_scaled", "merkle_2p24_tree"):
            assert sec[key]["hbm_bytes"] > 0


def test_bench_reads_the_instruction_count_from_the_record():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "83945" not in src and "83 945" not in src.split('"""', 2)[2], "instruction count hard-coded in bench.py"
    assert 'prof["valu_instructions_per_wave"]' in src
