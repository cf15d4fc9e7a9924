#!/usr/bin/env python3
"""Rebuild libcosim_hip.so with -Rpass-analysis=kernel-resource-usage and print one line per kernel.

    python tools/kres.py [flags]    # table: kernel, VGPRs, AGPRs, SGPR/VGPR spills, scratch B/lane, occupancy, LDS
                                    # (extra hipcc flags replace the product build's tuning flags, engine.HIPCC_TUNING)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from cosim_amd.engine import CSRC, HIPCC_TUNING, LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-o", LIB_PATH,
           os.path.join(CSRC, "cosim_engine.hip"), "-Rpass-analysis=kernel-resource-usage"] + (sys.argv[1:] if len(sys.argv) > 1 else HIPCC_TUNING)
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode:
        sys.stderr.write(p.stderr)
        sys.exit(p.returncode)
    rows, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line)
        if not m:
            m = re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:") or t.startswith("Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'sSpill':>6s} {'vSpill':>6s} {'scratch':>7s} {'occ':>3s} {'LDS':>6s}")
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = name.replace("cosim::", "").replace("(cosim::KArgs)", "")
        print(f"{name[:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('SGPRs Spill', '?'):>6s} "
              f"{r.get('VGPRs Spill', '?'):>6s} {r.get('ScratchSize [bytes/lane]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>3s} "
              f"{r.get('LDS Size [bytes/block]', '?'):>6s}")


if __name__ == "__main__":
    main()
