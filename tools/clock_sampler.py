"""Engine clock and power of the card while a command runs: what frequency the issue-bound figures of bench.py should be read at.

usage: clock_sampler.py <label> <out.json> -- <command ...>
The command runs as a child process; this process never touches the GPU. Every 100 ms it reads, for every card under
/sys/class/drm, hwmon freq1_input (engine clock, Hz), power1_average / power1_input (microwatts), gpu_busy_percent and the line
of pp_dpm_sclk marked '*'. Cards that never report busy are dropped from the summary (a box shows the host's other cards too).
When sysfs has none of these, `rocm-smi --showclocks --showpower --json` is polled once a second instead.
Prints one JSON line: per busy card, mean / p10 / min / max engine clock over the samples taken while the card was busy, mean power.
"""
import glob, json, os, subprocess, sys, time


def rd(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def cards():
    out = []
    for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        if "-" in os.path.basename(c):
            continue
        dev = os.path.join(c, "device")
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        out.append((os.path.basename(c), dev, hw[0] if hw else None))
    return out


def sample(card):
    _, dev, hw = card
    s = {}
    if hw:
        v = rd(os.path.join(hw, "freq1_input"))
        if v and v.strip().isdigit():
            s["sclk_mhz"] = int(v) / 1e6
        for name in ("power1_average", "power1_input"):
            v = rd(os.path.join(hw, name))
            if v and v.strip().isdigit():
                s["power_w"] = int(v) / 1e6
                break
    v = rd(os.path.join(dev, "gpu_busy_percent"))
    if v and v.strip().isdigit():
        s["busy"] = int(v)
    v = rd(os.path.join(dev, "pp_dpm_sclk"))
    if v:
        for line in v.splitlines():
            if "*" in line:
                try:
                    s["dpm_sclk_mhz"] = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                except (IndexError, ValueError):
                    pass
    return s


def smi_sample():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showuse", "--json"], capture_output=True, text=True, timeout=10)
        return json.loads(r.stdout)
    except Exception as e:   # noqa: BLE001 - a sampler must not end the run it watches
        return {"error": str(e)}


def stats(xs):
    xs = sorted(xs)
    if not xs:
        return None
    return {"n": len(xs), "mean": round(sum(xs) / len(xs), 1), "min": round(xs[0], 1), "p10": round(xs[len(xs) // 10], 1), "max": round(xs[-1], 1)}


def main():
    label, out = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    cs = cards()
    have_sysfs = any(sample(c) for c in cs)
    child = subprocess.Popen(cmd)
    series = {c[0]: [] for c in cs}
    smi = []
    t0 = time.time()
    while child.poll() is None:
        if have_sysfs:
            for c in cs:
                s = sample(c)
                if s:
                    s["t"] = round(time.time() - t0, 2)
                    series[c[0]].append(s)
            time.sleep(0.1)
        else:
            smi.append({"t": round(time.time() - t0, 2), "smi": smi_sample()})
            time.sleep(1.0)
    summary = {"label": label, "command": " ".join(cmd), "rc": child.returncode, "seconds": round(time.time() - t0, 1), "source": "sysfs" if have_sysfs else "rocm-smi", "cards": {}}
    for name, ss in series.items():
        busy = [s for s in ss if s.get("busy", 0) >= 50]
        if not busy:
            continue
        key = "sclk_mhz" if any("sclk_mhz" in s for s in busy) else "dpm_sclk_mhz"
        summary["cards"][name] = {"clock_field": key, "sclk_mhz_while_busy": stats([s[key] for s in busy if key in s]),
                                  "power_w_while_busy": stats([s["power_w"] for s in busy if "power_w" in s]),
                                  "samples": len(ss), "busy_samples": len(busy)}
    if not have_sysfs:
        summary["smi_first"] = smi[:2]
        summary["smi_mid"] = smi[len(smi) // 2:len(smi) // 2 + 2]
    with open(out, "w") as f:
        json.dump({"summary": summary, "series": {k: v[::5] for k, v in series.items() if k in summary["cards"]}, "smi": smi}, f)
    print(json.dumps(summary))
    sys.exit(child.returncode)


if __name__ == "__main__":
    main()
