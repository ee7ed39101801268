"""Fold the GetCost ceiling-probe session (python tools/diag_r4.py getcost > x.jsonl, with the DMVS_GC_EXP=4 builds of tools/build_variant.py
next to the product) into profiles/r5_getcost_ceiling_probe.json, the file bench.py quotes as roofline.ceiling_probe:

    python tools/ceiling_probe.py profiles/r5_getcost_ceiling_probe_b96.jsonl
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rows = [json.loads(ln) for ln in open(sys.argv[1]) if ln.startswith("{")]
    probes = sorted({r["build"] for r in rows if r["build"].startswith("gcexp4")})

    def us(geo, conf, build):
        return [r["us"] for r in rows if r["geometry"] == geo and r["conf"] == conf and r["build"] == build][0]

    ref_build = "gcsame" if any(r["build"] == "gcsame" for r in rows) else "product"      # the product source built like the probes (see diag_r4.py)

    def pair(geo, conf):
        return {"product_us": us(geo, conf, ref_build), "probe_us": min(us(geo, conf, b) for b in probes)}
    sys.path.insert(0, ROOT)
    import bench
    out = {"what": "GetCost ceiling probe (tools/diag_r4.py getcost; builds of warp_quad.hip with -DDMVS_GC_EXP=4: the product kernel's own projection, "
                   "texel masks, addresses and loads, the loaded registers only waited for -- no dot, no hat weights, no scatter; DMVS_QUAD_TPT = texels "
                   "in flight per trip; probe_us = the fastest of the probe builds)",
           "batch": rows[0]["B"], "shape": "cfg2 stage 2: 128 x 160 pixels, 5 source views, C = 32 fp32, 6 hypotheses",
           "algorithmic_bytes_per_launch": 1785200640, "kernel_source_sha": bench.kernel_source_hash(), "probe_builds": probes,
           "noise_no_confidence": pair("noise", None), "noise_random_confidence": pair("noise", "random"), "scene_confidence_0p5": pair("scene", 0.5),
           "gate_0p60_us": round(1785200640 / (0.6 * 8e12) * 1e6, 1), "rows": rows}
    with open(os.path.join(ROOT, "profiles", "r5_getcost_ceiling_probe.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("noise_no_confidence", "noise_random_confidence", "scene_confidence_0p5", "gate_0p60_us")}))


if __name__ == "__main__":
    main()
