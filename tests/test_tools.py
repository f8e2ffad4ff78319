"""CPU tests of the tuning / profiling tools (tools/): the in-frame trace join and the per-shape schedule pick are what
`live2diff_amd/igemm_tuned.json` and the numbers in profiles/ come from, so their logic is pinned on synthetic data."""
import csv
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(path, rows):
    with open(path, "w") as f:
        f.write("idx,kernel,dims,flops,bytes,dispatches\n")
        for i, (k, dims, fl, by, nd) in enumerate(rows):
            f.write(f"{i},{k},{dims},{fl},{by},{nd}\n")


def test_frame_trace_joins_dispatches_with_the_plan(tmp_path):
    plan = [("igemm_kernel", "taps1 M8 N8 K8 Kp64 s1 u0 e0 b1 S2 t2 v1 o0", 2e6, 1e3, 2),      # split-K: GEMM + epilogue
            ("layernorm_kernel", "rows8 C64", 0, 4e3, 1),
            ("copy", "", 0, 0, 0),                                                            # memcpy: no kernel dispatch
            ("tattn_stream_kernel", "N2 T8 C64 L16", 1e3, 8e3, 1),
            ("rowgemm_kernel", "M64 N64 K64 e0 p1 w1 t1 m1 tr0", 5e5, 2e4, 1),                  # round-3 kernel families
            ("pconv_kernel", "B1 H8 W16 C64 N64 patch8x16 o0", 9e6, 3e4, 1)]
    _plan(tmp_path / "plan.csv", plan)
    db = sqlite3.connect(tmp_path / "t.db")
    db.execute("create table kernels (name text, start integer, end integer)")
    names = ["void igemm_kernel<64, 64, 0, 64, 3>(IGemmArgs)", "igemm_splitk_epilogue(IGemmArgs, int)",
             "layernorm_kernel(...)", "void tattn_stream_ringlw_kernel<5, 16, 4, 5>(TAttnArgs, ...)",
             "void rowgemm_kernel<1, 1, 24, 0, 512>(RowGemmArgs)", "void pconv_kernel<8, 16>(PConvArgs)"]
    t = 1000
    db.execute("insert into kernels values ('some_torch_kernel', 0, 10)")           # not a product kernel: ignored
    for frame in range(4):                                                            # frames 0, 1 are skipped as warm-up
        for j, n in enumerate(names):
            dur = (10_000, 3_000, 5_000, 20_000, 7_000, 30_000)[j] + (frame >= 2) * 1_000
            db.execute("insert into kernels values (?, ?, ?)", (n, t, t + dur))
            t += dur + 500                                                            # 0.5 us gap after every dispatch
    db.commit()
    db.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "frame_trace.py"), str(tmp_path / "t.db"),
                          str(tmp_path / "plan.csv"), str(tmp_path / "out.csv"), "2"], capture_output=True, text=True, check=True)
    assert "2 frames matched, 6 dispatches/frame" in out.stdout
    rows = list(csv.DictReader(open(tmp_path / "out.csv")))
    assert [r["kernel"] for r in rows] == [p[0] for p in plan]
    assert abs(float(rows[0]["us_in_frame"]) - 15.0) < 1e-6          # GEMM 11 us + its split-K epilogue 4 us
    assert abs(float(rows[1]["us_in_frame"]) - 6.0) < 1e-6 and float(rows[2]["us_in_frame"]) == 0.0
    assert abs(float(rows[3]["us_in_frame"]) - 21.0) < 1e-6 and abs(float(rows[0]["gap_us"]) - 1.0) < 1e-6
    assert abs(float(rows[4]["us_in_frame"]) - 8.0) < 1e-6 and abs(float(rows[5]["us_in_frame"]) - 31.0) < 1e-6


def test_igemm_pick_keeps_the_baseline_unless_clearly_better(tmp_path):
    def trace(path, cfgs):
        with open(path, "w") as f:
            f.write("idx,kernel,dims,us_in_frame,gap_us,tflops,gbps\n")
            for i, (m, s, t, v, us) in enumerate(cfgs):
                f.write(f"{i},igemm_kernel,taps1 M{m} N64 K64 Kp64 s1 u0 e0 b1 S{s} t{t} v{v} o0,{us},0,1,1\n")
            f.write("99,layernorm_kernel,rows8 C64,5.0,0,0,1\n")
    trace(tmp_path / "trace_base.csv", [(128, 1, 2, 1, 10.0), (128, 1, 2, 1, 12.0), (512, 1, 2, 1, 20.0)])
    trace(tmp_path / "trace_1_2_5.csv", [(128, 2, 1, 5, 8.0), (128, 2, 1, 5, 8.0), (512, 2, 1, 5, 19.8)])   # -27 % / -1 %
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "igemm_pick.py"), str(tmp_path / "t.json"),
                          str(tmp_path / "trace_base.csv"), str(tmp_path / "trace_1_2_5.csv")], capture_output=True, text=True, check=True)
    shapes = json.load(open(tmp_path / "t.json"))["shapes"]
    assert shapes["1,128,64,64,0,1"] == [1, 2, 5]                    # clearly better: switched
    assert shapes["1,512,64,64,0,1"] == [2, 1, 1]                    # within 3 %: the baseline's pick stays
    assert "igemm in-frame" in out.stdout
