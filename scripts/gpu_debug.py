"""Run the HIP path on the golden cases and print per-tap errors (no asserts) - used while bringing kernels up."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import torch
from helpers import CASES, load_case, weights_for, get_ctor
from fcvsr_amd.arch import CVSR_freq as A

prec = os.environ.get("FCVSR_PRECISION", "f32")
cases = sys.argv[1:] or CASES
for name in cases:
    x, gold, meta = load_case(name)
    model = get_ctor(meta["ctor"])(**meta["kwargs"])
    model.load_state_dict(weights_for(meta), strict=True)
    model = model.to("cuda")
    with torch.no_grad():
        y = model(x.cuda())            # builds the engine
        model._engine.taps = {}
        t = time.time(); y = model(x.cuda()); torch.cuda.synchronize(); dt = time.time() - t
    taps = model._engine.taps
    mse = float(((y.cpu().double() - gold["out"].double()) * 255).pow(2).mean())
    import math
    print(f"== {name} [{prec}]: {dt*1e3:.1f} ms; out finite={bool(torch.isfinite(y).all())}; "
          f"PSNR(build, reference) = {20*math.log10(255/math.sqrt(mse)) if mse > 0 else float('inf'):.2f} dB")
    for k, g in gold.items():
        if k not in taps:
            print(f"   {k:24s} (no tap)"); continue
        t_ = taps[k].cpu()
        if t_.shape != g.shape:
            print(f"   {k:24s} SHAPE {tuple(t_.shape)} vs {tuple(g.shape)}"); continue
        err = float((t_ - g).abs().max()); sc = float(g.abs().max())
        print(f"   {k:24s} max|err|={err:.3e}  max|ref|={sc:.3e}  rel={err/max(sc,1e-30):.2e}")
