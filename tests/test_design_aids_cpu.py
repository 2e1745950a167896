"""The design aids that closed the decoder's table layout keep working: tests/tools/sim_hierarchy.c at 1/16 of an XCD's share
(224 streams, 256 KiB of L2) reproduces the counters the full-size run was checked against (profiles/r06_line_utilisation_sim.txt)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(layouts, extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "sim_hierarchy.py"), "--streams", "224", "--layouts", layouts, "--jobs", "4", "--extra", extra],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    for m in re.finditer(r"RESULT layout=(\d+) config=(\d) rdreq=([\d.]+) wrreq=([\d.]+) fabric=([\d.]+) l2hit=([\d.]+)", r.stdout):
        out[int(m.group(1))] = tuple(float(m.group(i)) for i in range(3, 7))
    return out


def test_table_layout_simulator_reproduces_the_measured_counters_and_its_verdict():
    res = _run("0,1,5,10", "l2_kb=256 mall_kb=2048 wt=1")
    rd, wr, fabric, hit = res[0]
    # deployed layout, measured on hardware (profiles/r05_simple_summary.txt): 0.68 fills + 1.18 write requests per decoded byte, 64 % L2 hits
    assert 0.62 < rd < 0.74 and 1.08 < wr < 1.25 and 0.60 < hit < 0.70, res[0]
    # numeric byte order: x 1.23 fetch on hardware (profiles/r03c_byte_rank_layout_same_box.txt)
    assert 1.15 < res[1][0] / rd < 1.40
    # the verdict: dense first-touch packing sends no fewer requests than the deployed layout, and the clairvoyant bound is nowhere near -30 %
    assert res[5][2] > 0.99 * fabric
    assert 0.90 * fabric < res[10][2] < fabric
    # a write-back L2 would have coalesced a third of the write requests away -- the model that did NOT match the counters
    wb = _run("0", "l2_kb=256 mall_kb=2048 wt=0")
    assert wb[0][1] < 0.8 * wr
