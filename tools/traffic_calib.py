"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU for the access patterns of the quad warp kernels.

    # on the GPU box, counters in their own passes (gpurun refuses --pmc together with the trace domains):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dirF> -- python tools/traffic_calib.py run
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <dirW> -- python tools/traffic_calib.py run
    python tools/traffic_calib.py reduce <dirF> <dirW>  > profiles/r3_traffic_calibration.json

`run` launches kernels that move a KNOWN number of bytes through HBM (2 GiB buffers: 8x the 256 MiB Infinity Cache, every
byte touched exactly once per launch); `reduce` divides the known bytes by what the counters reported (unit: KiB per
dispatch, as tools/pmc_traffic.py reads them) -> the factor to multiply a raw counter with, per pattern."""
import ctypes
import glob
import csv
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BYTES = 2 << 30
# kernel name substring -> (pattern label, which counter it calibrates, known bytes per launch)
PATTERNS = {
    "calib_stream_read": ("stream_read_16B_per_lane", "FETCH_SIZE", BYTES),
    "calib_quad_gather": ("quad_gather_64B_units", "FETCH_SIZE", None),       # whole texel and first-unit-only launches: see run()
    "calib_store_runs": ("store_4B_per_lane_64B_runs", "WRITE_SIZE", BYTES),
    "calib_stream_write": ("stream_write_16B_per_lane", "WRITE_SIZE", BYTES),
}


def build():
    so = "/tmp/dmvs_calib.so"
    src = os.path.join(HERE, "calib", "calib_kernels.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", so], check=True)
    return ctypes.CDLL(so)


def run():
    import torch
    lib = build()
    lib.calib_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    buf = torch.full((BYTES // 4,), 7, dtype=torch.int32, device="cuda:0")
    sink = torch.zeros(1 << 16, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    # order matters for `reduce`: the gather kernel runs 3 x whole texels (arg 2), then 3 x first 64-byte unit only (arg 1)
    for which, arg, reps in ((1, 0, 3), (2, 2, 3), (2, 1, 3), (3, 8, 3), (4, 0, 3)):
        for _ in range(reps):
            rc = lib.calib_run(which, buf.data_ptr(), sink.data_ptr(), BYTES, arg, st)
            assert rc == 0, rc
            torch.cuda.synchronize()
    print("calibration kernels done")


def counters(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "calib_" in r["Kernel_Name"]:
                out.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"]) * 1024.0))
    return {k: [v for _, v in sorted(vs)] for k, vs in out.items()}


def reduce_(fetch_dir, write_dir):
    fe, wr = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
    res = {"buffer_bytes": BYTES, "unit": "factor = known bytes / (raw counter x 1024 B)", "patterns": {}}

    def add(label, known, vals):
        m = sum(vals) / len(vals)
        res["patterns"][label] = {"known_bytes": known, "counter_bytes_mean": int(m), "factor": round(known / m, 4), "launches": len(vals)}
    for k, v in fe.items():
        if "calib_stream_read" in k:
            add("FETCH_SIZE stream read, 16 B per lane coalesced", BYTES, v)
        elif "calib_quad_gather" in k:
            add("FETCH_SIZE quad gather, whole 128-byte texels in scrambled order (64 B per quad and load)", BYTES, v[:3])
            add("FETCH_SIZE quad gather, first 64 bytes of every 128-byte line only", BYTES // 2, v[3:])
    for k, v in wr.items():
        if "calib_store_runs" in k:
            add("WRITE_SIZE 4 B per lane, 64-byte runs in 4 planes per wave", BYTES, v)
        elif "calib_stream_write" in k:
            add("WRITE_SIZE stream write, 16 B per lane coalesced", BYTES, v)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        reduce_(sys.argv[2], sys.argv[3])
