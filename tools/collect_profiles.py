"""Copy the judged rocprofv3 summaries from gpurun_out/<run> into profiles/<round> and rebuild
profiles/traffic.json from the PMC summaries (tools/refresh_profiles.sh produces <run> on the GPU box).
    python tools/collect_profiles.py gpurun_out/r01b profiles/r01
"""
import glob, hashlib, json, os, re, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = "pothoscomms_amd/csrc/"
# the kernel each workload's PMC run measured and the source files it is built from: bench.py reports `traffic` only while
# these files still hash to what they did when the measurement was filed (run this script on the tree the GPU run used)
KERNELS = {
    "fir255": ("fir_cf32_ols4096_kernel", [C + "fir_ols.hip", C + "fft4096.hpp", C + "pcx_sched.hpp"]),
    "fmchain": ("fmchain_cf32_ols4096_kernel", [C + "fir_ols.hip", C + "fft4096.hpp", C + "pcx_sched.hpp"]),
    "fft4096": ("fft_r16_kernel", [C + "fft_r16.hip", C + "fft4096.hpp"]),
    "direct255": ("fir_cf32_direct_kernel", [C + "fir_direct.hip"]),
    "decim8": ("fir_cf32_ols4096_decim_batched_kernel", [C + "fir_ols_decim.hip", C + "fft4096.hpp", C + "pcx_sched.hpp"]),
    "interp4": ("fir_cf32_ols4096_interp_batched_kernel", [C + "fir_ols_decim.hip", C + "fft4096.hpp", C + "pcx_sched.hpp"]),
    "fir255_i16": ("fir_cf64_ip_kernel", [C + "fir_ols_f64.hip", C + "fft_f64.hpp", C + "pcx_sched.hpp"]),
    "fir4097": ("fir_cf32_upols_kernel", [C + "fir_ols_part.hip", C + "fft4096.hpp"]),
    "fir8193": ("fir_cf32_upols_kernel", [C + "fir_ols_part.hip", C + "fft4096.hpp"]),
    "rotate": ("map_kernel", [C + "elementwise.hip", C + "vec_io.hpp"]),
}
def hashes(files):
    return {f: hashlib.sha256(open(os.path.join(ROOT, f), "rb").read()).hexdigest()[:16] for f in files}
os.makedirs(dst, exist_ok=True)

def newest(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None

for name in ("bench_default.json", "bench_other_workloads.jsonl", "bench_driver_flags.json", "bench_two_ranks_one_gpu_gloo.json",
             "f64_lab.txt", "ip64_parts.txt", "ab_ip64_sched.txt", "ab_upols.txt", "rccl_group_lab.txt", "pytest_gpu.txt", "soak.txt",
             "ab_sched.txt", "ab_oversub.txt", "ab_fft4096_family.txt", "ab_fft_family_rounds.txt", "shard_probe.txt", "host_path.txt", "two_blocks.txt", "chain_path.txt", "pcie_lab.txt", "ubench_roofs.txt",
             "ols_lab_summary.txt", "sweep_fir_taps.txt", "sweep_elementwise.txt", "sweep_fft_sizes.txt", "sweep_fft_f64.txt",
             "sweep_fft_mixed.txt", "sweep_fir_f64.txt", "real_f32_fir.txt", "ols_lab3_summary.txt", "transient_probe.txt", "shard4_trace.txt",
             "sweep_fft_q15_large.txt", "bench_native_two_shards_one_gpu.json", "bench_native_two_shards_one_gpu_fmchain.json",
             "bench_native_c3_eight_shards_one_gpu.json", "bench_native_c3_eight_shards_one_gpu_fmchain.json", "bench_eight_ranks_one_gpu_gloo.json",
             "bench_native_c3_eight_shards_one_gpu_double_buffered.json", "bench_native_two_shards_one_gpu_double_buffered.json",
             "clk_lab.txt", "shard_probe_c3.txt", "floor_table.txt", "chain_path.txt", "c_pcie_probe.txt", "shard_probe_threads.txt",
             "bench_native_c3_eight_shards_one_gpu_submit_threads.json") + tuple(
                 "bench_rccl_rank_rehearsal%s%s.json" % (a, b) for a in ("", "_unpipelined", "_plain_same_box") for b in ("", "_fmchain")):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
f = newest(os.path.join(src, "bench_kt", "**", "*_kernel_stats.csv"))
if f:
    shutil.copy(f, os.path.join(dst, "bench_fir255_kernel_stats.csv"))
traffic = {}
for wl in KERNELS:
    summ = os.path.join(src, wl, "summary.txt")
    if not os.path.exists(summ):
        continue
    shutil.copy(summ, os.path.join(dst, wl + "_rocprofv3_summary.txt"))
    f = newest(os.path.join(src, wl, "kt", "**", "*_kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join(dst, wl + "_kernel_stats.csv"))
    txt = open(summ).read()
    fetch = re.search(r"FETCH_SIZE\s+([0-9.e+]+)", txt)
    write = re.search(r"WRITE_SIZE\s+([0-9.e+]+)", txt)
    counters = {m.group(1): float(m.group(2)) for m in re.finditer(r"^pmc\s+(\S+)\s+([0-9.e+]+)", txt, re.M)
                if m.group(1) in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_INSTS_SALU",
                                  "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE")}
    if fetch and write:
        fk, wk = float(fetch.group(1)), float(write.group(1))
        traffic[wl] = {
            "counters": counters,
            "kernel": KERNELS[wl][0],
            "sources": hashes(KERNELS[wl][1]),
            "hbm_bytes_per_launch": int((2 * fk + wk) * 1024),
            "FETCH_SIZE_KiB_raw": fk,
            "WRITE_SIZE_KiB": wk,
            "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams -> doubled "
                          "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
            "source": "%s/%s_rocprofv3_summary.txt (separate --pmc passes, mean of 3 launches)" % (dst, wl),
        }
old = {}
tj = os.path.join(os.path.dirname(dst.rstrip("/")), "traffic.json")
if os.path.exists(tj):
    old = json.load(open(tj))
old.update(traffic)
json.dump(old, open(tj, "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in old.items()}))
