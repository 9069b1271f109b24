#!/usr/bin/env python3
"""Stamp the counter record a GPU session wrote (gpurun_out/prof_<tag>/hbm_traffic.json, by tools/summarize_profile.py ON
the GPU box, which has no .git) with the commit whose sources were measured, and install it as profiles/hbm_traffic.json.

    python tools/stamp_profile.py gpurun_out/prof_r5/hbm_traffic.json [profiles/r5]

The record is keyed by hashes of the kernel sources as they were ON THE BOX.  The stamp is only written when those hashes
equal the hashes of the sources of HEAD -- computed from `git show HEAD:<file>`, not from the working tree -- so that the
commit named in the record is one whose tree really is the measured kernel.  Nothing in the record is ever edited by hand:
a kernel change means a new measurement (tests/test_profile_record.py enforces both)."""
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hades252_amd import build  # noqa: E402

CSRC_REL = "hades252_amd/csrc"


def _show(commit: str, rel: str) -> bytes:
    return subprocess.run(["git", "show", "%s:%s" % (commit, rel)], cwd=ROOT, check=True, capture_output=True).stdout


def perm_fast_hash_at(commit: str) -> str:
    """build.perm_fast_hash() of the tree of `commit` (same recipe, file contents from git)."""
    h = hashlib.sha256(" ".join(build.FLAGS).encode())
    for d in build.PERM_FAST_DEPS:
        h.update(d.encode() + b"\0" + _show(commit, CSRC_REL + "/" + d))
    keep = False
    for line in _show(commit, CSRC_REL + "/hades_constants.inc").decode().splitlines(keepends=True):
        if line.startswith("#define "):
            keep = line.split()[1] in build.PERM_FAST_TABLES
        if keep:
            h.update(line.encode())
            keep = line.rstrip().endswith("\\")
    return h.hexdigest()


def device_source_hash_at(commit: str) -> str:
    h = hashlib.sha256(" ".join(build.FLAGS).encode())
    for d in sorted(build.DEVICE_DEPS + build.LAUNCH_POLICY_DEPS):
        h.update(d.encode() + b"\0" + _show(commit, CSRC_REL + "/" + d))
    return h.hexdigest()


def main():
    src = sys.argv[1]
    rec = json.load(open(src))
    head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, check=True, capture_output=True, text=True).stdout.strip()
    if "hash_note" in rec or "measured_at_commit" in rec:
        raise SystemExit("refusing: the record already carries a stamp or a note -- start from the GPU box's file")
    if rec.get("kernel_source_hash") != perm_fast_hash_at(head):
        raise SystemExit("refusing: the measured kernel (%s) is not HEAD's (%s) -- commit the measured sources first, or "
                         "measure again" % (str(rec.get("kernel_source_hash"))[:12], perm_fast_hash_at(head)[:12]))
    rec["measured_at_commit"] = head
    sec = rec.get("secondary_kernels")
    if isinstance(sec, dict):
        if sec.get("device_source_hash") == device_source_hash_at(head):
            sec["measured_at_commit"] = head
        else:
            print("secondary_kernels: device sources differ from HEAD's -- dropped")
            del rec["secondary_kernels"]
    dst = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    with open(dst, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
    if len(sys.argv) > 2:
        os.makedirs(sys.argv[2], exist_ok=True)
        shutil.copy(dst, os.path.join(sys.argv[2], "hbm_traffic.json"))
    print("stamped with", head, "->", dst)


if __name__ == "__main__":
    main()
