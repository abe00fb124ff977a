#!/usr/bin/env python
"""Host-side profile of Kandinsky2_1HIP.generate_text2img at C2 (developer tool): cProfile of three warm calls, top cumulative entries, plus
wall time with a device synchronisation after every phase.  Run on the GPU box: python tools/prof_e2e_host.py"""
import argparse, cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import kandinsky2_amd as k22

a = argparse.Namespace(chains=1, sched_steps=50, bs=1, size=768)
dev = torch.device("cuda:0")
arch = k22.make_arch(k22.MODEL_CONFIG_2_1, inpainting=False)
pipe = bench._seeded_pipeline(a, arch, None, dev, torch.bfloat16)
prompt = "a red cat, 4k photo"
gen = lambda: pipe.generate_text2img(prompt, num_steps=50, batch_size=1, guidance_scale=4, h=768, w=768, sampler="p_sampler",
                                     prior_cf_scale=4, prior_steps="25", output_type="tensor")
gen(); torch.cuda.synchronize()
gen(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    gen()
torch.cuda.synchronize()
print(f"generate_text2img: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    gen()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
