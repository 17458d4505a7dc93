#!/usr/bin/env python3
"""The 'numbers of record' table of DESIGN.md from ONE call of tools/round_profile.sh: usage tools/numbers_of_record.py gpurun_out/<tag> <tag>.
Prints the markdown rows; the round-over-round remarks in them are this round's (edit here when a round ends)."""
import json, re, sys
S, TAG = sys.argv[1].rstrip("/") + "/", sys.argv[2]
J = lambda f: json.loads(open(S + f).read().strip().splitlines()[-1])
b, bs = J("bench.json"), J("bench_serial.json")
r, rr, cpu = b["roofline"], b["roofline_random_rays"], b["cpu_baseline"]
vp = json.load(open(S + "valu_profile.json"))
drv = [l.split() for l in open(S + "driver_cmd.txt") if l.startswith("driver command")]
dv = [float(x[8]) for x in drv]; dms = [float(x[9]) for x in drv]; dfr = [float(x[11]) for x in drv]
cfg = [json.loads(l) for l in open(S + "configs.jsonl")]
def c(sub): return next(d for d in cfg if sub in d.get("config", ""))
reh = open(S + "multi_gpu_rehearsal.txt").read()
def ms(world): return [float(m) for m in re.findall(r"rehearse-world %d \(20 steps[^\n]*\n\s+[0-9.]+ Mrays/s ([0-9.]+) ms/step" % world, reh)]
one = float(re.search(r"1 GPU, 20 steps:\n\s+[0-9.]+ Mrays/s ([0-9.]+) ms/step", reh).group(1))
k4 = float(re.search(r"'ms_per_step': ([0-9.]+)", reh).group(1))
rng = lambda v, f="%.4f": (f % min(v)) if min(v) == max(v) else (f % min(v) + " – " + f % max(v))
n = lambda v: "{:,.0f}".format(v)
mk = r["main_kernel_counters"]
vx = c("drop-in vx_")
g2 = c("leaf_max 2")
tests = re.search(r"(\d+) passed", open(S + "pytest_gpu.txt").read()).group(1)
rows = [
 ("headline, default run (%d steps; 5 frames per set of launches, 2 sets in flight)" % b["steps"], "**%s Mrays/s**, %.4f ms per step (round 4: 10,772)" % (n(b["value"]), b["ms_per_step"]), "`%s_bench.json`" % TAG),
 ("the driver's command (`--gpus 1 --steps 20 --warmup 5`), five runs", "**%s – %s Mrays/s** (%s ms), `roofline.frac` %s (round 4: 10,446 – 10,549)" % (n(min(dv)), n(max(dv)), rng(dms, "%.3f"), rng(dfr, "%.3f")), "`%s_driver_cmd.txt`" % TAG),
 ("serial frames (`--frames-in-flight 1`)", "%s Mrays/s, %.4f ms" % (n(bs["value"]), bs["ms_per_step"]), "`%s_bench_serial.json`" % TAG),
 ("roofline, dominant kernel (`rt_persistent_kernel<JOB_RENDER_SHADOW>` + EXACT + shade)",
  "%s wave64 VALU instructions per frame ÷ %.4f ms = %.1f G/s against 1,024 SIMDs × %.4f GHz held ÷ 2 = %s G/s: **frac %.3f**; lane utilisation %.3f, %.2f of wave cycles in `s_waitcnt`; **HBM traffic %.0f MB per frame (round 4: 373), of it %.0f MB written (171)** = %.2f TB/s against 5.13 GB algorithmic in the reference's order (1,236 B/ray)"
  % ("{:,}".format(vp["valu_instr_per_frame"]), r["kernel_ms"], r["achieved"], r["clock_ghz_held"], "{:,.1f}".format(r["peak"]), r["frac"], mk["lane_utilisation"], mk["wait_any_of_wave_cycles"],
     vp["hbm_bytes_per_frame"] / 1e6, vp["hbm_write_bytes"] / 1e6, vp["hbm_bytes_per_frame"] / r["kernel_ms"] / 1e9), "`valu_profile.json` ← `%s_pmc.txt`" % TAG),
 ("random rays (16 Mi, `vxrt_trace`)", "%s Mrays/s, %.3f ms per launch; %.2f G instructions per launch: **frac %.3f**, lane utilisation %.2f, wait %.2f; %.1f B/ray algorithmic = %.2f TB/s = %.2f of the HBM peak as north_star words it"
  % (n(rr["mrays_s"]), rr["ms_per_launch"], rr["valu_instr_per_launch"] / 1e9, rr["frac"], rr["lane_utilisation"], rr["wait_any_of_wave_cycles"], rr["bytes"]["bytes_per_ray"], rr["bytes"]["algorithmic_GBs"] / 1e3, rr["bytes"]["frac_of_hbm_peak"]), "same, `%s_pmc_summary_random_rays.txt`" % TAG),
 ("CPU, reference object code on the box's host cores", "%.1f Mrays/s (%d threads, the frame's primary rays); the other legs in the line's `cpu_baseline`" % (cpu["value"], cpu["cores"]), "`%s_bench.json`" % TAG),
 ("3840×2160", "%s Mrays/s serial (%.3f ms)" % (n(c("configs[3] on one GPU")["mrays_s"]), c("configs[3] on one GPU")["ms_per_frame_serial"]), "`%s_configs.jsonl`, `extras.other_configs`" % TAG),
 ("bunny-class 1024²", "%s Mrays/s (%.3f ms)" % (n(c("configs[1]")["mrays_s"]), c("configs[1]")["ms_per_frame_serial"]), "same"),
 ("diffuse bounce", "%s Mrays/s serial, %s two frames in flight" % (n(c("diffuse bounce")["mrays_s"]), n(c("diffuse bounce")["mrays_s_2_in_flight"])), "same"),
 ("hairball 10 M triangles, 16 spp AO", "%s Mrays/s (%.2f ms); scene built on the host in %.1f s; **now a leg of the default 1-GPU bench line**" % (n(c("configs[4]")["mrays_s"]), c("configs[4]")["ms_per_frame"], c("configs[4]")["host_build_s"]), "same"),
 ("software twin, 1080p primary rays", "%s Mrays/s (%.3f ms); VALU frac 0.42, lanes 0.63, wait 0.59" % (n(c("software twin")["mrays_s"]), c("software twin")["ms_per_frame"]), "same, `%s_other_kernels_roofline.txt`" % TAG),
 ("GPU-built BLAS (`vxrt_bvh_build`, 1,048,576 triangles)", "built in **%.1f ms** (round 4: 1.5 ms without the reinsertion step); the headline frame on it, serial: **%s Mrays/s against %s on the CPU builder's tree = %.1f %%** (round 4: 7,950 = 90.5 %%), %.2f node fetches + %.2f triangle tests per ray against 18.71 + 3.65"
  % (g2["build_ms_median"], n(g2["mrays_s"]), n(g2["sah_tree_mrays_s"]), 100.0 * g2["mrays_s"] / g2["sah_tree_mrays_s"], g2["node_fetches_per_ray"], g2["tri_fetches_per_ray"]), "same (leg 8), `r05_g_gpu_reinsertion.txt`"),
 ("`vx_*` call sequence, C++ host", "%.3f ms per frame = **%s Mrays/s**; with `vx_copy_from_dev` %.3f ms = **%s**; **`-s 5` (five samples in one `vx_start`): %.3f ms per start = %s Mrays/s per traced ray (with the copy %s)**"
  % (vx["cxx_host_ms_per_frame_start_wait"], n(vx["cxx_host_mrays_s_start_wait"]), vx["cxx_host_ms_per_frame_with_copy_from_dev"], n(vx["cxx_host_mrays_s_with_copy"]), vx["cxx_host_spp5_ms_per_start_wait"], n(vx["cxx_host_spp5_mrays_s_start_wait"]), n(vx["cxx_host_spp5_mrays_s_with_copy"])), "same"),
 ("rehearsed rank-0 pipeline, 20-step runs", "%s / %s / %s ms per step at N = 2 / 4 / 8 (sets 9+9+2 / 16+4 / 12+8) against %.4f on one GPU = %.2f× / %.2f× / %.2f× before the network (means; boxes of the pool differ by ± 3 %% at N = 8: 0.0598 – 0.0636 over this round's passes); 3840×2160 at N = 8: %.4f ms per step"
  % (rng(ms(2)), rng(ms(4)), rng(ms(8)), one, one / (sum(ms(2)) / len(ms(2))), one / (sum(ms(4)) / len(ms(4))), one / (sum(ms(8)) / len(ms(8))), k4), "`%s_multi_gpu_rehearsal.txt`" % TAG),
]
print("(%s GPU tests green in the same call)" % tests)
for a_, b_, c_ in rows:
    print("| %s | %s | %s |" % (a_, b_, c_))
