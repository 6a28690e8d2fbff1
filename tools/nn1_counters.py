"""Derived metrics of the k = 1 kernels from the rocprofv3 --pmc passes tools/pmc_flat.sh left under
gpurun_out/pmcf_<mode>_<scene>/ (one pass per counter group, --kernel-trace only): instructions per wave, lanes per
VALU instruction, VALU busy fraction, occupancy, waiting share, L1 hit rate, LDS conflict share.
usage: nn1_counters.py <out.json> <label>=<dir>:<kernel substring> ..."""
import collections, csv, glob, json, sys


def load(d, kernel):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{d}/p*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in acc.items()}


def derive(c):
    g = c.get
    d = {}
    waves = g("SQ_WAVES")
    cycles = g("SQ_BUSY_CYCLES", 0) / 32.0  # 32 shader engines
    if waves:
        d["valu_instructions_per_wave"] = g("SQ_INSTS_VALU", 0) / waves
        d["salu_instructions_per_wave"] = g("SQ_INSTS_SALU", 0) / waves
        d["lds_instructions_per_wave"] = g("SQ_INSTS_LDS", 0) / waves
        d["vmem_read_instructions_per_wave"] = g("SQ_INSTS_VMEM_RD", 0) / waves
    if g("SQ_INSTS_VALU"):
        d["active_lanes_per_valu_instruction"] = g("SQ_THREAD_CYCLES_VALU", 0) / g("SQ_INSTS_VALU")
    if cycles:
        d["kernel_cycles"] = cycles
        d["valu_busy_fraction (ACTIVE_INST_VALU * 4 / 1024 SIMDs / kernel cycles)"] = g("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cycles
        d["waves_per_simd_average (WAVE_CYCLES * 4 / 1024 / kernel cycles)"] = g("SQ_WAVE_CYCLES", 0) * 4 / 1024 / cycles
    if g("SQ_WAVE_CYCLES"):
        d["wave_time_waiting_fraction (WAIT_ANY / WAVE_CYCLES)"] = g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES")
        d["wave_time_issue_stalled_fraction (WAIT_INST_ANY / WAVE_CYCLES)"] = g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
    if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
        d["l1_hit_rate"] = 1 - g("TCP_TCC_READ_REQ_sum", 0) / g("TCP_TOTAL_CACHE_ACCESSES_sum")
    if g("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_share_of_lds_cycles"] = g("SQ_LDS_BANK_CONFLICT", 0) / g("SQ_LDS_IDX_ACTIVE")
    return d


out = {}
for spec in sys.argv[2:]:
    label, rest = spec.split("=", 1)
    d, kernel = rest.rsplit(":", 1)
    c = load(d, kernel)
    out[label] = {"kernel": kernel, "derived": derive(c), "raw_per_launch": c}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in out.items():
    print(k, json.dumps(v["derived"], indent=1))
