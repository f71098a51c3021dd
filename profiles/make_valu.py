#!/usr/bin/env python3
"""VALU work of one batch from two rocprofv3 PMC passes (profiles/run_valu.sh) + the measured issue cost per instruction
class (profiles/ubench/valu_issue.hip) -> the VALU-issue roof of bench.py's roofline.valu block.
usage: make_valu.py <pmc_valu.txt> <pmc_mix.txt> <valu_issue.jsonl> <out.json>

Per kernel (dispatches are serialised by the counter collection: the kernel's own figures):
  valu_busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)   (SQ_ACTIVE_INST_* count quad-cycles)
Per batch (one batch = one frontend launch):
  valu_roof_ms = sum over classes of instructions x measured TIME per instruction and SIMD / 1024 SIMDs
  valu_busy_ms_counters = sum over kernels of SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz): the same roof from
                          the counters alone (SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU within 4 %: four cycles per instruction)
  issue_roof_ms = sum over kernels of SQ_ACTIVE_INST_ANY x 4 cycles / (1024 SIMDs x 2.4 GHz): the quad-cycles the batch's waves
                  spend with an instruction of ANY kind in flight (vector, scalar, memory, LDS, branch), spread over the SIMDs.
                  This is the roof the pipeline sits on: rounds 2 and 3 measured 7.97 / 8.1, 7.12 / 7.09 and 7.04 / 7.0 ms
                  (roof / batch period) -- the batch's kernels are chains of dependent instructions with one to three waves per
                  SIMD, which issue ONE instruction per SIMD and quad-cycle between them, whatever its kind
The time per instruction is the microbenchmark's kernel time / (instructions per wave x waves per SIMD), the minimum over
2, 4 and 8 waves per SIMD: 1.6-2.1 ns for the floating-point classes = 4 cycles of a 16-lane SIMD at 2.0-2.4 GHz, as the
data sheet has it.  (Round 3's first version multiplied the benchmark's s_memtime ticks per instruction by 1 / 2.4 GHz.
With 8 waves per SIMD the waves do not all run at once -- the kernel took twice as long as its slowest wave's ticks --,
so "1.4-2.6 cycles per instruction" and a roof of 1.79 ms were a factor two too low: the counters said so all along.)
"""
import json, re, sys

CLOCK_GHZ = 2.36  # measured under the pipeline's load (profiles/r05_clocks_power.txt, r05_device_activity.txt); rounds 1-4 assumed 2.4
N_SIMD = 1024


def parse(path):
    out = {}
    for ln in open(path):
        m = re.match(r"(.+?)\s+dispatches=(\d+)\s+(.*)", ln)
        if not m:
            continue
        name = m.group(1).strip().replace("void ", "").replace("tfrec::", "")
        d = {k: float(v) for k, v in (kv.split("=") for kv in m.group(3).split())}
        d["dispatches"] = int(m.group(2))
        out.setdefault(name, []).append(d)
    return out


valu, mix = parse(sys.argv[1]), parse(sys.argv[2])
cost = {}  # ns per instruction and SIMD
for ln in open(sys.argv[3]):
    ln = ln.strip()
    if ln.startswith("{"):
        j = json.loads(ln)
        if j["waves_per_simd"] >= 2 and "clock_ghz" in j:
            ns = j["cycles_per_inst_per_simd"] / j["clock_ghz"]  # = kernel time / (instructions per wave x waves per SIMD)
            cost[j["class"]] = min(cost.get(j["class"], 1e9), round(ns, 3))
cls_cost = {"fp64": cost["v_fma_f64"], "fma_f32": cost["v_pk_fma_f32"], "cvt": cost["v_cvt_f64_i32"], "int32": cost["v_add_u32"],
            "other": cost["v_fma_f32"], "salu": cost["s_add_u32"]}
nb = max(d["dispatches"] for n, ds in valu.items() if n.startswith("frontend_kernel") for d in ds)
kernels = {}
busy_valu = busy_sca = busy_any = 0.0
tot = dict(valu=0.0, fp64=0.0, fma_f32=0.0, cvt=0.0, int32=0.0, other=0.0, salu=0.0)
for name, ds in valu.items():
    if name.startswith("__"):
        continue
    for k, d in enumerate(ds):  # (names truncated to 34 characters: several template instances may share one)
        mx = mix.get(name, [{}] * len(ds))[k] if k < len(mix.get(name, [])) else {}
        per = d["dispatches"] / nb
        fp64 = mx.get("SQ_INSTS_VALU_FMA_F64", 0) + mx.get("SQ_INSTS_VALU_ADD_F64", 0) + mx.get("SQ_INSTS_VALU_MUL_F64", 0)
        f32 = mx.get("SQ_INSTS_VALU_FMA_F32", 0)
        cvt = mx.get("SQ_INSTS_VALU_CVT", 0)
        i32 = mx.get("SQ_INSTS_VALU_INT32", 0)
        nv = d.get("SQ_INSTS_VALU", 0)
        other = max(0.0, nv - fp64 - f32 - cvt - i32)
        gui_cycles = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
        key = name if len(ds) == 1 else "%s #%d" % (name, k)
        kernels[key] = dict(launches_per_batch=round(per, 2), insts_valu=nv, insts_salu=d.get("SQ_INSTS_SALU", 0), fp64=fp64,
                            fma_f32=f32, cvt=cvt, int32=i32, other=other, kernel_cycles_alone=gui_cycles,
                            kernel_ms_alone=round(gui_cycles / (CLOCK_GHZ * 1e6), 4),
                            valu_busy=round(4.0 * d.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, gui_cycles * N_SIMD), 4),
                            inst_active_ms=round(4.0 * d.get("SQ_ACTIVE_INST_ANY", 0) * per / (N_SIMD * CLOCK_GHZ * 1e6), 4))
        busy_valu += 4.0 * d.get("SQ_ACTIVE_INST_VALU", 0) * per
        busy_sca += 4.0 * d.get("SQ_ACTIVE_INST_SCA", 0) * per
        busy_any += 4.0 * d.get("SQ_ACTIVE_INST_ANY", 0) * per
        for c, v in (("valu", nv), ("fp64", fp64), ("fma_f32", f32), ("cvt", cvt), ("int32", i32), ("other", other),
                     ("salu", d.get("SQ_INSTS_SALU", 0))):
            tot[c] += v * per
issue = sum(tot[c] * cls_cost[c] for c in ("fp64", "fma_f32", "cvt", "int32", "other"))  # SIMD-ns
out = dict(note=__doc__.strip(), ns_per_instruction_per_simd=cls_cost, per_batch=tot, valu_issue_simd_ns_per_batch=issue,
           valu_roof_ms=round(issue / N_SIMD / 1e6, 4),
           valu_busy_ms_counters=round(busy_valu / (N_SIMD * CLOCK_GHZ * 1e6), 4),
           salu_roof_ms=round(tot["salu"] * cls_cost["salu"] / N_SIMD / 1e6, 4),
           salu_busy_ms_counters=round(busy_sca / (N_SIMD * CLOCK_GHZ * 1e6), 4),
           issue_roof_ms=round(busy_any / (N_SIMD * CLOCK_GHZ * 1e6), 4),
           sum_of_kernel_ms_alone=round(sum(k["kernel_ms_alone"] * k["launches_per_batch"] for k in kernels.values()), 3),
           kernels=kernels)
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("per_batch", "ns_per_instruction_per_simd", "valu_roof_ms", "valu_busy_ms_counters",
                                      "salu_roof_ms", "salu_busy_ms_counters", "issue_roof_ms", "sum_of_kernel_ms_alone")}, indent=1))
