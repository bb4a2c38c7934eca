#!/usr/bin/env python3
"""Development aid: board power / shader clock (sysfs hwmon) sampled while the ViT forward loops (gpurun only).
usage: vit_power.py [seconds] [idle|vit|gemm]"""
import sys, os, time, glob, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import torch
from hiptagsearch import synth
from hiptagsearch.tagger import ViTTagger
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
print("hwmon:", hw)
def rd(p):
    try: return open(p).read().strip()
    except Exception as e: return None
for h in hw:
    for f in ("power1_cap", "power1_cap_max", "power1_average", "power1_input", "freq1_input", "freq1_label", "temp1_input", "temp2_input"):
        print(h, f, rd(os.path.join(h, f)))
for c in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
    print(c, (rd(c) or "").replace("\n", " | "))
cfg = dict(synth.VIT_B16_448); w = synth.vit_weights(cfg, seed=0); B = 64
m = ViTTagger(cfg, w, max_batch=B)
imgs = torch.randint(0, 256, (B, 448, 448, 3), dtype=torch.uint8, device="cuda")
probs = torch.empty((B, cfg["num_classes"]), dtype=torch.float32, device="cuda")
samples = []; stop = [False]
def sampler():
    h = hw[0] if hw else None
    while not stop[0]:
        t = time.perf_counter()
        p = rd(os.path.join(h, "power1_average")) or rd(os.path.join(h, "power1_input"))
        f = rd(os.path.join(h, "freq1_input"))
        samples.append((t, int(p) / 1e6 if p else -1, int(f) / 1e6 if f else -1))
        time.sleep(0.05)
th = threading.Thread(target=sampler); th.start()
time.sleep(1.0)
t_start = time.perf_counter(); n = 0
while time.perf_counter() - t_start < secs:
    for _ in range(8): m.forward_u8(imgs, probs=probs, want="probs")
    torch.cuda.synchronize(); n += 8
t_end = time.perf_counter()
time.sleep(1.0); stop[0] = True; th.join()
print("forwards %d  %.2f ms each  %.0f images/s" % (n, (t_end - t_start) / n * 1e3, n * B / (t_end - t_start)))
for t, p, f in samples[::4]:
    print("t %+6.2f s  power %7.1f W  sclk %6.0f MHz" % (t - t_start, p, f))
