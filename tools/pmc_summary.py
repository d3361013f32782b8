#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of tools/profile_round.sh into the round's committed summary:
    python tools/pmc_summary.py r03_b > profiles/r03_b_pmc_summary.txt        (also rewrites profiles/pmc_traffic.json)
Reads gpurun_out/<tag>_pmc_<mode>_<group>/p_counter_collection.csv (mode = exact | fast | c4 | sched3), the kernel times
the same passes printed (gpurun_out/<tag>_pmc_<mode>_a.log) and the library build id of the in-tree .so.
Derived figures (MI355X_MICROARCH.md): a wave64 VALU instruction occupies its SIMD for 2 cycles; FETCH_SIZE tallies
128-byte requests at 64 bytes on gfx950 (doubled here; calibrated for coalesced reads only -- both figures are given);
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import ctypes
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, CUS = 1024, 256


def table(d):
    acc = collections.defaultdict(list)
    name = None
    with open(os.path.join(d, "p_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if "render_kernel" not in row["Kernel_Name"]:
                continue
            name = row["Kernel_Name"].replace("(crt_render_args)", "").replace("void ", "").replace("(anonymous namespace)::", "")
            acc[(row["Counter_Name"], row["Dispatch_Id"])].append(float(row["Counter_Value"]))
    per = collections.defaultdict(list)
    for (c, _), v in acc.items():
        per[c].append(sum(v))
    return {c: sum(v) / len(v) for c, v in per.items()}, name


def log_figures(path):
    ms = rays = None
    for line in open(path):
        m = re.search(r"rays (\d+) checksum", line)
        if m:
            rays = int(m.group(1))
        m = re.search(r"median\s+([0-9.]+) ms", line)
        if m:
            ms = float(m.group(1))
    return ms, rays


def main(tag):
    out = os.path.join(ROOT, "gpurun_out")
    lib = ctypes.CDLL(os.path.join(ROOT, "cuda_ray_tracer_amd", "libcrt_hip.so"))
    lib.crt_build_id.restype = ctypes.c_char_p
    build = lib.crt_build_id().decode()
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    print(f"# round-3 PMC passes ({tag}; rocprofv3 --pmc, one counter group per run, tools/profile_round.sh; this file: tools/pmc_summary.py)")
    print(f"# library build {build} (commit {commit}); C2 = tools/tune.py --spp 500 (1200x675, 500 spp, depth 50), C4 = the 1M-sphere scene at 1920x1080, 64 spp; two launches per pass")
    traffic = {"source": f"profiles/{tag}_pmc_summary.txt: rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes (tools/profile_round.sh), KB per dispatch",
               "correction": "MI355X_MICROARCH.md 'HBM': on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so fetch bytes are DOUBLED here "
                             "(calibrated for coalesced reads; the raw figure is kept beside it); WRITE_SIZE kept raw (uncalibrated for 64-bit integer atomics)"}
    rows = []
    for mode, key, label in (("exact", "c2", "C2 exact"), ("fast", "c2_fast", "C2 fast"), ("sched3", None, "C2 exact, SCHED=3 (walker/owner)"),
                             ("c4", "c4", "C4 exact (scene in HBM)")):
        t, names = {}, set()
        for g in "abcdefgh":
            d = os.path.join(out, f"{tag}_pmc_{mode}_{g}")
            if os.path.isdir(d):
                tt, nm = table(d)
                t.update(tt)
                names.add(nm)
        if not t:
            continue
        ms, rays = log_figures(os.path.join(out, f"{tag}_pmc_{mode}_a.log"))
        d = {"mode": label, "kernel": sorted(n for n in names if n)[0], "ms": ms, "rays": rays}
        if "GRBM_GUI_ACTIVE" in t:
            ms_e, _ = log_figures(os.path.join(out, f"{tag}_pmc_{mode}_e.log"))
            d["clock_GHz"] = t["GRBM_GUI_ACTIVE"] / 8 / (ms_e * 1e-3) / 1e9
        clock = d.get("clock_GHz", 2.33)
        cyc = ms * 1e-3 * clock * 1e9
        d["valu_per_ray"] = t["SQ_INSTS_VALU"] / rays
        d["lane_util"] = t["SQ_THREAD_CYCLES_VALU"] / (64 * t["SQ_ACTIVE_INST_VALU"])
        d["valu_busy"] = 2 * t["SQ_INSTS_VALU"] / (SIMDS * cyc)
        if "SQ_INSTS_SALU" in t:
            d["salu_per_ray"] = t["SQ_INSTS_SALU"] / rays
        if "SQ_INSTS_VALU_FMA_F32" in t:
            d["add_mul_fma_per_ray"] = [t[f"SQ_INSTS_VALU_{k}_F32"] / rays for k in ("ADD", "MUL", "FMA")]
            d["wave_issue_stall"] = t["SQ_WAIT_INST_ANY"] / t["SQ_WAVE_CYCLES"]
            d["wave_active"] = t["SQ_ACTIVE_INST_ANY"] / t["SQ_WAVE_CYCLES"]
        if "SQ_WAIT_ANY" in t:
            d["wave_waitcnt"] = t["SQ_WAIT_ANY"] / t["SQ_WAVE_CYCLES"]
        if "SQ_LDS_IDX_ACTIVE" in t:
            d["lds_busy"] = t["SQ_LDS_IDX_ACTIVE"] / (CUS * cyc)
            d["lds_conflict_share"] = t["SQ_LDS_BANK_CONFLICT"] / t["SQ_LDS_IDX_ACTIVE"]
            d["wave_lds_issue_stall"] = t["SQ_WAIT_INST_LDS"] / t["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in t:
            d["fetch_raw_B"], d["write_B"] = t["FETCH_SIZE"] * 1024, t.get("WRITE_SIZE", 0) * 1024
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in t:
            d["l1_accesses_per_clk_per_cu"] = t["TCP_TOTAL_CACHE_ACCESSES_sum"] / (CUS * cyc)
            d["l1_hit"] = 1 - t["TCP_TCC_READ_REQ_sum"] / t["TCP_TOTAL_CACHE_ACCESSES_sum"]
        if "TCC_HIT_sum" in t:
            d["l2_hit"] = t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"])
            d["l2_miss_B_at_64"] = t["TCC_MISS_sum"] * 64
        rows.append((d, t))
        if key:
            e = {"build_id": build, "commit": commit, "width": 1920 if mode == "c4" else 1200, "spp": 64 if mode == "c4" else 500,
                 "kernel": "render_kernel<WORLD=BVH, IN_LDS=%d, COUNT=0, BLOCK=768, SCHED=2>%s" % (0 if mode == "c4" else 1, " [arith=fast]" if mode == "fast" else ""),
                 "profiled_symbol": d["kernel"], "fetch_bytes_raw": int(d["fetch_raw_B"]), "fetch_bytes_per_launch": int(2 * d["fetch_raw_B"]),
                 "write_bytes_per_launch": int(d["write_B"]), "file": f"profiles/{tag}_pmc_summary.txt",
                 "valu": {"valu_instructions_per_ray": round(d["valu_per_ray"], 2), "lane_utilisation": round(d["lane_util"], 3),
                          "valu_issue_busy_at_2_cycles": round(d["valu_busy"], 3), "shader_clock_GHz": round(clock, 3),
                          "source": f"profiles/{tag}_pmc_summary.txt (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU), GRBM_GUI_ACTIVE)"}}
            if mode == "c4":
                e["note"] = ("memory-side (fabric) read requests of the L2s, Infinity-Cache hits included; the 99 MB scene fits the 256 MB "
                             "Infinity Cache, so HBM itself sees less")
                e["l1_accesses_per_clk_per_cu"] = round(d["l1_accesses_per_clk_per_cu"], 3)
                e["l1_hit"], e["l2_hit"] = round(d["l1_hit"], 3), round(d["l2_hit"], 3)
                e["wave_cycles_waiting_on_memory"] = round(d["wave_waitcnt"], 3)
            else:
                e["note"] = "scene is LDS/L1-resident; writes = 64-bit atomic adds of the per-unit tile flushes; algorithmic frame bytes 19.4 MB"
            traffic[key] = e
    f = lambda v, p=1: "-" if v is None else (f"{v * 100:.{p}f} %")
    print("# derived:")
    for d, _ in rows:
        print(f"#   {d['mode']}: {d['kernel']}")
        print(f"#     kernel time under the PMC profiler {d['ms']:.2f} ms, {d['rays']} rays per launch" + (f", shader clock {d['clock_GHz']:.3f} GHz" if "clock_GHz" in d else ""))
        print(f"#     VALU wave-instructions per ray {d['valu_per_ray']:.2f}" + (f", SALU {d['salu_per_ray']:.2f}" if "salu_per_ray" in d else "")
              + f"; lane utilisation of VALU instructions {f(d['lane_util'])} (SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)); "
              f"VALU issue busy {f(d['valu_busy'])} (2 cycles per wave64 instruction)")
        if "add_mul_fma_per_ray" in d:
            a, m, fm = d["add_mul_fma_per_ray"]
            print(f"#     fp32 add / mul / fma per ray {a:.1f} / {m:.1f} / {fm:.1f}; wave cycles: issuing {f(d['wave_active'])}, waiting for an issue slot "
                  f"{f(d['wave_issue_stall'])}, in s_waitcnt {f(d.get('wave_waitcnt'))}")
        elif "wave_waitcnt" in d:
            print(f"#     wave cycles in s_waitcnt {f(d['wave_waitcnt'])}")
        if "lds_busy" in d:
            print(f"#     LDS array busy {f(d['lds_busy'])} of the kernel's cycles, {f(d['lds_conflict_share'])} of that bank conflicts; wave cycles stalled at "
                  f"LDS issue {f(d['wave_lds_issue_stall'], 2)}")
        if "fetch_raw_B" in d:
            sec = d["ms"] * 1e-3
            print(f"#     memory side per launch: FETCH_SIZE {d['fetch_raw_B'] / 1e6:.2f} MB raw (x2 gfx950 correction = {2 * d['fetch_raw_B'] / 1e6:.2f} MB), "
                  f"WRITE_SIZE {d['write_B'] / 1e6:.2f} MB = {d['fetch_raw_B'] / sec / 1e9:.3g}-{2 * d['fetch_raw_B'] / sec / 1e9:.3g} GB/s read, {d['write_B'] / sec / 1e9:.1f} GB/s written")
        if "l1_hit" in d:
            print(f"#     L1 (TCP): {d['l1_accesses_per_clk_per_cu']:.2f} cache accesses per clock per CU, hit rate {f(d['l1_hit'])}; L2 (TCC) hit rate "
                  f"{f(d.get('l2_hit'))}, misses x 64 B = {d.get('l2_miss_B_at_64', 0) / 1e9:.0f} GB per launch")
    for d, t in rows:
        print(f"# raw counters, mean per dispatch -- {d['mode']}")
        for c in sorted(t):
            print(f"{c:32s} {t[c]:.6g}")
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
