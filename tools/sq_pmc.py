#!/usr/bin/env python3
"""SQ counters of one kernel, per launch: `python tools/sq_pmc.py <kernel regex> -- <command...>` runs the command under
rocprofv3 --pmc (two passes: wave-state cycles, instruction mix; never combined with tracing) and prints the mean per
launch of every counter for the kernels whose name matches.  Units: SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves (MI355X_MICROARCH.md, rocprofv3 PMC slots)."""
import csv, glob, os, re, shutil, subprocess, sys, tempfile

PASSES = [["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"], ["SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES"],
          ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"],
          ["GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_SALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_BRANCH"]]


def main():
    pat = re.compile(sys.argv[1])
    cmd = sys.argv[sys.argv.index("--") + 1:]
    res = {}
    for counters in PASSES:
        d = tempfile.mkdtemp(prefix="sqpmc_", dir="/tmp")
        try:
            r = subprocess.run(["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", *cmd],
                               env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=600)
            if r.returncode != 0:
                print("pass failed:", counters, r.stderr[-400:])
                continue
            n0 = sum(len(v) for v in res.values())
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if pat.search(row["Kernel_Name"]):
                        res.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            if sum(len(v) for v in res.values()) == n0:
                print("pass gave no rows:", counters, r.stderr[-300:])
        finally:
            shutil.rmtree(d, ignore_errors=True)
    for k, v in res.items():
        print(f"{k:28s} launches={len(v):3d} mean={sum(v) / len(v):.4g}")


if __name__ == "__main__":
    main()
