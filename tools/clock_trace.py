"""Clock / power trace of the GPU through a sustained bench run (the evidence behind "the chip clocks to its power budget"):
starts `python bench.py --steps N ...` and samples the shader clock, the socket power and the temperature while it runs.

    python tools/clock_trace.py OUT.md [steps=200] [extra bench.py flags ...]

Sampler: the amdgpu hwmon / sysfs files when they exist (cheap: ~1 ms per sample, 20 ms period), else `rocm-smi --showclocks --showpower --json`
(one process per sample, ~0.3 s period).  The table is bucketed by phase: idle before the run, start-up, the timed region (detected by the power rising
above idle + 25 % of the range), the tail."""
import glob
import json
import os
import subprocess
import sys
import time


def sysfs_sources():
    """hwmon files of EVERY amdgpu card of the node (a box shows all of the node's cards in sysfs, not only the one the container may use):
    all are sampled, the card whose power moves with the run is the one reported"""
    cards = []
    for dev in sorted(glob.glob('/sys/class/drm/card*/device')):
        hw = sorted(glob.glob(os.path.join(dev, 'hwmon', 'hwmon*')))
        if not hw:
            continue
        h, out = hw[0], {'card': dev.split('/')[-2]}
        for key, names in (('power_uw', ('power1_average', 'power1_input')), ('sclk_hz', ('freq1_input',)), ('mclk_hz', ('freq2_input',)),
                           ('temp_mc', ('temp2_input', 'temp1_input'))):
            for n in names:
                p = os.path.join(h, n)
                if os.path.exists(p):
                    out[key] = p
                    break
        if 'power_uw' in out and 'sclk_hz' in out:
            cards.append(out)
    return cards


def visible_pci_bus():
    """PCI address (0000:bb:dd.f) of HIP device 0 of this container"""
    try:
        r = subprocess.run([sys.executable, '-c', 'import torch; p = torch.cuda.get_device_properties(0); '
                            'print("%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))'], capture_output=True, text=True, timeout=120)
        out = r.stdout.strip().splitlines()
        return out[-1] if out else None
    except Exception:                                           # noqa: BLE001
        return None


def read_int(path):
    try:
        return int(open(path).read().strip())
    except Exception:                                           # noqa: BLE001
        return None


def smi_sample():
    try:
        r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--json'], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        card = d[sorted(d)[0]]
        val = {}
        for k, v in card.items():
            kl = k.lower()
            try:
                if 'sclk' in kl and 'clock' in kl:
                    val['sclk_mhz'] = float(str(v).strip('()MHz mhz'))
                elif 'power' in kl and ('socket' in kl or 'average' in kl or 'current' in kl):
                    val['power_w'] = float(v)
                elif 'temperature' in kl and ('hotspot' in kl or 'junction' in kl):
                    val['temp_c'] = float(v)
            except ValueError:
                pass
        return val
    except Exception:                                           # noqa: BLE001
        return {}


def main():
    out_path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    extra = sys.argv[3:]
    cards = sysfs_sources()
    use_sysfs = bool(cards)
    period = 0.02 if use_sysfs else 0.0

    def sample():
        if use_sysfs:
            out = []
            for src in cards:
                v = {'power_w': (read_int(src['power_uw']) or 0) / 1e6, 'sclk_mhz': (read_int(src['sclk_hz']) or 0) / 1e6}
                if 'temp_mc' in src:
                    v['temp_c'] = (read_int(src['temp_mc']) or 0) / 1e3
                out.append(v)
            return out
        return [smi_sample()]

    samples = []
    t0 = time.time()
    for _ in range(10 if use_sysfs else 3):                     # idle baseline
        samples.append((time.time() - t0, 'idle', sample()))
        time.sleep(period)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--steps', str(steps), '--warmup', '5', '--no-cpu-baseline', '--no-conv-events'] + extra
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    while proc.poll() is None:
        samples.append((time.time() - t0, 'run', sample()))
        time.sleep(period)
    log = proc.stdout.read()
    line = [l for l in log.splitlines() if l.startswith('{')]
    bench = json.loads(line[-1]) if line else {}
    for _ in range(10 if use_sysfs else 2):
        samples.append((time.time() - t0, 'after', sample()))
        time.sleep(period)

    # the card under test: the one whose PCI address is the visible HIP device's (the node's other cards belong to other jobs and move too);
    # fallback: the card whose power moved most through the run
    ncard = len(samples[0][2])
    spread = []
    for c in range(ncard):
        pwc = [x[2][c].get('power_w') or 0.0 for x in samples]
        spread.append(max(pwc) - min(pwc))
    pick = max(range(ncard), key=lambda c: spread[c])
    bus = visible_pci_bus()
    how = 'largest power spread'
    if use_sysfs and bus:
        for c, src_ in enumerate(cards):
            if os.path.basename(os.path.realpath(os.path.join('/sys/class/drm', src_['card'], 'device'))).lower() == bus.lower():
                pick, how = c, f'PCI {bus} = the visible HIP device'
    samples = [(t, ph, v[pick]) for t, ph, v in samples]
    src = cards[pick] if use_sysfs else {}
    smi = smi_sample()
    pw = [s[2].get('power_w') for s in samples if s[2].get('power_w')]
    lo, hi = (min(pw), max(pw)) if pw else (0.0, 0.0)
    thr = lo + 0.5 * (hi - lo)
    busy = [s for s in samples if s[1] == 'run' and (s[2].get('power_w') or 0) >= thr]

    def stats(rows, key):
        v = sorted(x[2][key] for x in rows if x[2].get(key))
        if not v:
            return 'n/a'
        return f'min {v[0]:.0f}  median {v[len(v) // 2]:.0f}  max {v[-1]:.0f}'

    with open(out_path, 'w') as f:
        f.write(f'# clock / power trace of `bench.py --steps {steps} {" ".join(extra)}`\n\n')
        f.write(f'sampler: {"amdgpu hwmon sysfs, 20 ms period, " + str(ncard) + " cards sampled, power spread per card " + str([round(x) for x in spread]) + " W; picked by " + how + " -> " + json.dumps(src) if use_sysfs else "rocm-smi --showclocks --showpower --json, one process per sample"}\n\n')
        f.write(f'rocm-smi (the visible device, idle after the run): {json.dumps(smi)}\n\n')
        f.write(f'bench line: {json.dumps({k: bench.get(k) for k in ("value", "unit", "ms_per_step", "steps")})}\n\n')
        f.write(f'{len(samples)} samples over {samples[-1][0]:.1f} s; {len(busy)} of them under load (power >= {thr:.0f} W)\n\n')
        f.write('| phase | samples | shader clock MHz | socket power W | temperature C |\n|---|---|---|---|---|\n')
        for name, rows in (('idle before', [s for s in samples if s[1] == 'idle']), ('run, start-up (below the load threshold)',
                           [s for s in samples if s[1] == 'run' and s not in busy]), ('run, under load', busy),
                           ('after', [s for s in samples if s[1] == 'after'])):
            f.write(f'| {name} | {len(rows)} | {stats(rows, "sclk_mhz")} | {stats(rows, "power_w")} | {stats(rows, "temp_c")} |\n')
        f.write('\nunder-load samples in time order (t s, MHz, W):\n\n```\n')
        step = max(1, len(busy) // 60)
        for s in busy[::step]:
            f.write(f'{s[0]:7.2f}  {s[2].get("sclk_mhz", 0):6.0f}  {s[2].get("power_w", 0):6.0f}\n')
        f.write('```\n')
    print(open(out_path).read()[:1500])


if __name__ == '__main__':
    main()
