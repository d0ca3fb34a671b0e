"""The profile post-processing scripts run on data shaped like rocprofv3's (no GPU): scripts/sq_counters.py on a two-kernel counter collection."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLS = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name", "Workgroup_Size",
        "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp",
        "End_Timestamp"]


def _write(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=COLS)
        w.writeheader()
        for kernel, counter, value in rows:
            w.writerow({c: 0 for c in COLS} | {"Kernel_Name": kernel, "Counter_Name": counter, "Counter_Value": value})


def test_sq_counter_table(tmp_path):
    ka = "void migan::sepconv_pipe_kernel<0, 64, 64, true, false, 2, 4>(migan::SepArgs)"
    kb = "void migan::sepconv_pipedown_kernel<128, 64, 2, 12, 4>(migan::SepArgs)"
    sets = [("SQ_INSTS_VALU", 2.0e8), ("SQ_INSTS_MFMA", 8.0e6), ("SQ_INSTS_LDS", 4.0e7), ("SQ_VALU_MFMA_BUSY_CYCLES", 2.0e8),
            ("SQ_LDS_IDX_ACTIVE", 2.0e8), ("SQ_LDS_BANK_CONFLICT", 2.0e7), ("SQ_WAIT_INST_ANY", 4.0e8), ("SQ_WAVE_CYCLES", 1.0e9),
            ("SQ_ACTIVE_INST_ANY", 3.0e8)]
    for i, (c, v) in enumerate(sets):
        _write(tmp_path / f"cc_{i}.csv", [(ka, c, v), (ka, c, v), (kb, c, v / 2)])          # two launches of ka: averaged
    layers = [{"layer": "encoder.b512.conv1", "kernel": "migan::sepconv_pipe_kernel<0, 64, 64, true, false, 2, 4>", "ms": 1.0},
              {"layer": "encoder.b512.conv2.dwfir", "kernel": "migan::sepconv_pipedown_kernel<128, 64, 2, 12, 4>", "ms": 0.005},
              {"layer": "encoder.b512.conv2", "kernel": "migan::sepconv_pipedown_kernel<128, 64, 2, 12, 4>", "ms": 0.5}]
    json.dump(layers, open(tmp_path / "per_launch.json", "w"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sq_counters.py"), str(tmp_path)], check=True, capture_output=True, text=True).stdout
    rows = [l for l in out.split("\n") if l.startswith("| `")]
    assert len(rows) == 2 and "sepconv_pipe_kernel" in rows[0] and "(e.b512.conv1)" in rows[0]          # the longer kernel first
    cells = [c.strip() for c in rows[0].split("|")]
    # 2e8 VALU instructions over 1024 SIMDs in 1 ms at 2 GHz: 195 per SIMD and microsecond, one per 10.2 cycles; wait 40 %, conflicts 10 %
    assert cells[3] == "1.000" and cells[7] == "195" and cells[8] == "10.2" and cells[11] == "10%" and cells[12] == "40%" and cells[13] == "30%"
    assert "(e.b512.conv2)" in rows[1] and "| 0.500 |" in rows[1]                                     # the placeholder row of the fused launch is not averaged in


def test_isa_wait_scan_tokens():
    """scripts/isa_wait_scan.py: runs of loads / stores / LDS-DMAs are counted, waits keep their count, barriers are marked"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_wait_scan", os.path.join(ROOT, "scripts", "isa_wait_scan.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    body = ["global_load_dwordx4 v[0:3], v[4:5], off", "global_load_dword v6, v[4:5], off", "s_waitcnt vmcnt(1)", "v_add_f32 v0, v0, v1",
            "global_store_dwordx4 v[4:5], v[0:3], off nt", "buffer_load_dword v0, s[8:11], 0 offen lds", "buffer_load_dword v0, s[8:11], 0 offen lds",
            "s_waitcnt vmcnt(0)", "s_barrier", "s_waitcnt lgkmcnt(0)", "buffer_load_dwordx4 v[0:3], v9, s[8:11], 0 offen"]
    assert m.tokens(body) == "L2 w1 S1 D2 w0 | L1"
