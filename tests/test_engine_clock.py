"""bench.py's EngineClock: the card's engine clock and board power read from sysfs hwmon beside a leg. On a CPU box there is no
card: a made-up hwmon tree stands in, and a context without sysfs yields no summary (a run never fails for it)."""
import importlib.util
import os
import time


def _bench():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class _Ctx:
    def __init__(self, bdf):
        self.bdf = bdf

    def pci_bus_id(self):
        if self.bdf is None:
            raise RuntimeError("no device")
        return self.bdf


def test_reads_clock_and_power_of_the_card(tmp_path):
    hw = tmp_path / "0000:05:00.0" / "hwmon" / "hwmon7"
    hw.mkdir(parents=True)
    (hw / "freq1_input").write_text("2262000000\n")
    (hw / "power1_average").write_text("1208000000\n")
    b = _bench()
    clk = b.EngineClock(_Ctx("0000:05:00.0"), sysfs_root=str(tmp_path))
    assert clk.freq and clk.power
    with clk:
        time.sleep(0.15)
        (hw / "freq1_input").write_text("2100000000\n")
        time.sleep(0.15)
    s = clk.summary(skip_s=0.0)
    assert s["max"] == 2262.0 and s["min"] == 2100.0 and 2100.0 < s["mean"] < 2262.0
    assert s["board_power_w_mean"] == 1208.0 and s["samples"] >= 4 and s["peak_used_for_bounds"] == 2400.0
    late = clk.summary(skip_s=0.2)            # the ramp is left out: only the later clock remains
    assert late["max"] == 2100.0


def test_no_sysfs_no_summary(tmp_path):
    b = _bench()
    for ctx in (_Ctx(None), _Ctx("0000:ff:00.0")):
        clk = b.EngineClock(ctx, sysfs_root=str(tmp_path))
        assert clk.freq is None
        with clk:
            pass
        assert clk.summary() is None


def test_bench_help_prints():
    """`python bench.py --help` (argparse expands every help string with %): exits 0 without touching a GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    assert "--gpus" in r.stdout and "--steps" in r.stdout
