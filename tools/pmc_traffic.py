"""HBM-side traffic of the conv_gemm family from two rocprofv3 PMC passes over bench.py (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, FETCH_SIZE doubled on gfx950, both in KB).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/<round>_pmc_traffic.json
"""
import csv
import glob
import json
import os
import sys

# launches of the family in one headline forward (bench.py: roofline.launches_per_step): 184 aptp_conv_gemm since round 4 (the fused
# tail is off by default; rounds 2-3: 169 + 5 aptp_ff_tail = 174); overridable as the 4th argument
LAUNCHES_PER_STEP = int(sys.argv[4]) if len(sys.argv) > 4 else 184


def family_sum(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    total, launches = 0.0, 0
    seen = set()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            if "conv_gemm" in name or "lin_gemm" in name or "splitk_reduce" in name or "ff_tail" in name:
                total += float(r["Counter_Value"])
                if "conv_gemm" in name or "lin_gemm" in name or "ff_tail" in name:
                    key = (r.get("Dispatch_Id"), f)
                    if key not in seen:
                        seen.add(key)
                        launches += 1
    return total, launches


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch_kb, n_f = family_sum(fetch_dir, "FETCH_SIZE")
    write_kb, n_w = family_sum(write_dir, "WRITE_SIZE")
    steps_f, steps_w = n_f / LAUNCHES_PER_STEP, n_w / LAUNCHES_PER_STEP
    fetch_per_step = fetch_kb / steps_f
    write_per_step = write_kb / steps_w
    hbm = (2.0 * fetch_per_step + write_per_step) * 1024.0
    res = {
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph",
        "kernel_family": "conv_gemm_kernel / conv_gemm_dma_kernel / lin_gemm_kernel / ff_tail_kernel (+ splitk_reduce_kernel)",
        "forwards_profiled": steps_f,
        "FETCH_SIZE_KB_per_step_raw": fetch_per_step,
        "WRITE_SIZE_KB_per_step": write_per_step,
        "gfx950_correction": "FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads: doubled (MI355X_MICROARCH.md §HBM); WRITE_SIZE exact",
        "hbm_side_bytes_per_step": hbm,
        "launches_per_step": LAUNCHES_PER_STEP,
        "hbm_side_bytes_per_launch": hbm / LAUNCHES_PER_STEP,
        "note": "memory-side (fabric) requests of the L2; Infinity-Cache hits are counted, so this is traffic beyond L2, not DRAM traffic. "
                "Algorithmic bytes of the family per step ~ 2.9 GB (weights 1.0 GB once + activations in/out): the 3x3 taps re-read the input through L2.",
    }
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
