#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of one csrc/*.hip file, from hipcc's own resource remarks
(-Rpass-analysis=kernel-resource-usage).  usage: tools/kernel_resources.py mg_spot.hip [name-filter] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "endless-memory-gym_amd", "csrc")


def main():  # noqa: C901
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c",
           os.path.join(CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    out = r.stderr
    if r.returncode != 0:  # e.g. the backend's "Illegal instruction detected" when a spill reload lands on an odd register pair
        print("COMPILE FAILED:\n" + "\n".join(l for l in out.splitlines() if "error" in l.lower())[:2000])
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k in ("Function Name", "Name"):
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
        elif cur is not None:
            cur[{"SGPRs Spill": "sspill", "VGPRs Spill": "vspill"}.get(k, k.split(" ")[0])] = v
    print("%-100s %5s %5s %5s %7s %6s %6s %4s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "sspill", "vspill", "occ", "lds"))
    for r in rows:
        if flt and flt not in r["name"]:
            continue
        print("%-100s %5s %5s %5s %7s %6s %6s %4s %6s" % (r["name"][:100], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
                                                         r.get("ScratchSize"), r.get("sspill"), r.get("vspill"), r.get("Occupancy"), r.get("LDS")))


if __name__ == "__main__":
    main()
