"""One screen of the figures of a bench.py JSON line:  python tools/print_bench.py <bench.json>"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"value {d['value']:.4g} {d['unit']}  ms/step {d['ms_per_step']:.3f}  frac {r['frac']:.3f}  kernel_ms {r['kernel_ms']:.3f}  traffic {r['traffic']}")
def g(k, *path):
    x = d.get(k)
    for p in path:
        if not isinstance(x, dict) or "error" in x:
            return x.get("error") if isinstance(x, dict) else None
        x = x.get(p)
    return x
print("hbm K1 full ms", g("hbm_kernels", "K1_full_output", "ms"), "K3 by alloc", g("hbm_kernels", "K3_row_quantiles", "by_allocation_ms"), "rho0 K1", g("hbm_kernels_rho0", "K1_full_output", "ms"))
print("numpy ms", g("numpy_stream", "ms"), "class_api s", g("class_api_1e7", "seconds"), "kernels", g("class_api_1e7", "kernel_seconds"))
print("s60", g("s60", "paths_per_s"), "s60 ranged", g("s60_data_ranged", "paths_per_s"))
print("search s", g("search", "search_seconds"), "final run s", g("search", "final_run_seconds"), "months", g("search", "months_found"), "probes", g("search", "probes"))
cb = d.get("cpu_baseline", {})
print("cpu", {k: (round(v["value"]), v["cores"]) for k, v in cb.items() if isinstance(v, dict)}, "quota", cb.get("cgroup_cpu_quota_cores"))
print("accuracy", g("accuracy_10k", "abs_error"), g("accuracy_10k", "flipped_success_flags"))
