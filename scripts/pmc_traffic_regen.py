"""Regenerate profiles/pmc_traffic.json from the CURRENT kernels: for every BASELINE workload
two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate, as MI355X_MICROARCH.md prescribes) over the whole layer
step (scripts/layer_one.py), gfx950 correction FETCH_SIZE x 2, KB = 1024 B -- bench.measure_step_traffic.  The stage keys
(fwd_transform ...) are what bench.py falls back to when it cannot read the counters itself (--no-pmc, multi-GPU lines).
Usage (on the GPU box): python scripts/pmc_traffic_regen.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                            "profiles", "pmc_traffic.json")
METHOD = ("round 4: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over scripts/layer_one.py "
          "(the whole layer step, 2 steps: every kernel with its in-step cache state), bench.measure_step_traffic; FETCH_SIZE x 2 "
          "(gfx950 counts 128-B requests at 64 B), WRITE_SIZE x 1, KB = 1024 B; bytes per launch; `ms` = mean duration under the "
          "counter pass (ranking only)")
JOBS = [("fno2d_256_m64_c64_b32", "fno2d_256_m64_c64_b32", "f32", "dense"),
        ("fno2d_256_m64_c64_b32_bf16io", "fno2d_256_m64_c64_b32", "bf16", "dense"),
        ("fno2d_256_m64_c64_b32_tucker01", "fno2d_256_m64_c64_b32", "f32", "tucker"),
        ("fno3d_128_m32_c32_b8", "fno3d_128_m32_c32_b8", "f32", "dense"),
        ("fno2d_1024_m256_c128_b4", "fno2d_1024_m256_c128_b4", "f32", "dense"),
        ("darcy_421_m32_c32_b16", "darcy_421_m32_c32_b16", "f32", "dense")]
out = {}
for key, wl, io, kind in JOBS:
    got, note = bench.measure_step_traffic(bench.WORKLOADS[wl], io=io, kind=kind, timeout_s=240)
    if got is None:
        out[key] = {"_method": METHOD, "_error": note}
        print(key, "FAILED", note)
        continue
    ks = got["kernels"]
    ent = {"_method": METHOD, "step_traffic_B": got["step_traffic_B"], "kernels": ks}

    def total(*subs):
        return int(sum(v["traffic_B"] * v["launches_per_step"] for k, v in ks.items() if any(s in k for s in subs)))
    if kind == "dense":
        # a transform stage = every launch of its kernels in one step / the number of transforms of that type (2)
        fwd = total("k_fft2d_fwd3", "k_fft2d_fwd_mx", "k_f2p_r2c", "k_f2p_col_fwd", "k_pl128_fwd", "k_ax128<-1>") // 2
        inv = total("k_fft2d_inv3", "k_fft2d_inv_mx", "k_f2p_c2r", "k_f2p_col_inv", "k_pl128_inv", "k_ax128<1>") // 2
        ent.update(fwd_transform=fwd, adj_c2r_transform=fwd, inv_transform=inv, adj_r2c_transform=inv)
        cf = [v["traffic_B"] for k, v in ks.items() if k.startswith(("k_modegemm_dma<", "k_modegemm_sb<", "k_modegemm<"))]
        cb = [v["traffic_B"] for k, v in ks.items() if k.startswith(("k_modegemm_dma_bwd", "k_modegemm_sb_bwd"))]
        if cf:
            ent["contract_fwd"] = int(min(cf))
        if cb:
            ent["contract_bwd"] = int(cb[0])
    out[key] = ent
    print(key, got["step_traffic_B"], {k: v["traffic_B"] for k, v in ks.items()})
json.dump(out, open(OUT, "w"), indent=1)
print("wrote", OUT)
