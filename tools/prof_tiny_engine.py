import sys, os, time, cProfile, pstats, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import kandinsky2_amd as k22
arch = k22.make_arch(k22.tiny_model_config())
t0=time.time(); sd = k22.init_unet_state_dict(arch, seed=0); print("init sd", time.time()-t0)
B,h,w=2,16,16
full, pooled, image = k22.make_conditioning(arch, B, seed=2)
x=torch.randn(B,4,h,w); t=torch.tensor([10.0, 500.0])
def run(backend):
    t0=time.time()
    m = k22.Text2ImUNetHIP(arch, backend_dtype=backend, use_graph=False); t1=time.time()
    m.load_state_dict(sd); t2=time.time()
    m = m.to("cuda").eval(); torch.cuda.synchronize(); t3=time.time()
    o = m(x.cuda(), t.cuda(), full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda()); torch.cuda.synchronize(); t4=time.time()
    o = m(x.cuda(), t.cuda(), full_emb=full.cuda(), pooled_emb=pooled.cuda(), image_emb=image.cuda()); torch.cuda.synchronize(); t5=time.time()
    print(backend, f"ctor {t1-t0:.2f} load_sd {t2-t1:.2f} to_cuda {t3-t2:.2f} fwd1 {t4-t3:.2f} fwd2 {t5-t4:.3f}")
run(torch.float32); run(torch.float32); run(torch.bfloat16); run(torch.bfloat16)
pr=cProfile.Profile(); pr.enable(); run(torch.float32); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
