import sys, os, time
sys.path.insert(0, ".")
import numpy as np, torch
from oracle import caption_oracle as co
from image_caption_amd.engine import CaptionConfig, CaptionEngine
dev = torch.device("cuda", 0)
def run(dims, B, L, T, seed, peaked):
    p = (co.make_peaked_params if peaked else co.make_params)(seed, **dims)
    p = co.perturb_biases(p, seed + 1)
    img = co.make_images(seed + 2, B, L, dims["F"])
    ref = co.predict_image_caption_sequence(p, img, 1, T, 3, 4)
    cfg = CaptionConfig(B=B, W=1, L=L, F=dims["F"], D=dims["D"], E=dims["E"], H=dims["H"], U=dims["H"], V=dims["V"], T=T)
    eng = CaptionEngine(cfg, p, device=dev)
    tok, sc = eng.decode(torch.from_numpy(img).to(dev))
    s = torch.cuda.current_stream()
    t0 = time.time()
    while not s.query():
        if time.time() - t0 > 20: print("NOT FINISHED"); os._exit(3)
        time.sleep(0.01)
    got = tok.cpu().numpy()
    print(dims, "B", B, "invalid", eng.decode_invalid(), "equal", (got == ref).all(), "mismatch rows", int((got != ref).any(axis=1).sum()), flush=True)
    if not (got == ref).all():
        print(got[:3], ref[:3])
    eng.close()
run(dict(F=48, D=128, E=32, H=128, V=200), 5, 6, 6, 3, True)
run(dict(F=48, D=128, E=32, H=128, V=200), 40, 6, 6, 4, False)
run(dict(F=64, D=256, E=64, H=256, V=1000), 33, 9, 8, 5, True)
run(dict(F=2048, D=256, E=256, H=512, V=10000), 32, 64, 20, 0, True)
