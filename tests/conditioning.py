"""Test infrastructure: how ill-conditioned is a pixel of the reference's arithmetic?

Band / far pairs land far outside a part's box and are EXTRAPOLATED by the encoder with trilinear weights of 1e3..1e9 (DESIGN.md
§3): a one-ulp difference anywhere upstream moves such a pixel by 1e-4 and more — in the reference's own fp32 run as much as in the
kernels.  One fp32 run of the oracle is a single noisy sample of that (and changes with the host's thread count); so the noise scale of
a pixel is estimated as the largest move of the FLOAT64 result under several (default 4) independent fp32-ulp-sized random perturbations of the
rays, together with the deviation of the oracle's fp32 run.  Checker only."""
import torch


def pixel_noise(O, model64, b64, exact, n_samples, chunk=64, ref32=None, trials=4, seed=0, rerun32=None):
    """exact: (n,3) float64 rgb_map of the unperturbed float64 run -> (n,) noise scale per pixel.
    rerun32(chunk) -> (n,3): the oracle's fp32 render again — called under several thread counts and chunk sizes: other reduction
    orders, i.e. more samples of the fp32 arithmetic's INTERNAL rounding noise, which perturbing the rays alone does not reach (a pixel
    of the bench frame deviated 2.8e-5 on one host and > 1.7e-4 on others, same inputs, while the four ray perturbations moved it by
    less: round 5)."""
    g = torch.Generator().manual_seed(1000 + seed)
    noise = torch.zeros(exact.shape[0], dtype=torch.float64)
    if ref32 is not None:
        noise = torch.maximum(noise, (ref32.double() - exact).abs().max(1)[0])
    if rerun32 is not None:
        nt = torch.get_num_threads()
        try:
            for threads, ch in ((1, chunk), (2, max(chunk // 4, 1)), (5, max(chunk // 2, 1)), (max(nt // 2, 1), chunk * 2)):
                torch.set_num_threads(threads)
                with torch.no_grad():
                    noise = torch.maximum(noise, (rerun32(ch).double() - exact).abs().max(1)[0])
        finally:
            torch.set_num_threads(nt)
    sgn = lambda t: (torch.randint(0, 2, t.shape, generator=g).double() * 2.0 - 1.0)
    for _ in range(trials):
        bp = dict(b64)
        bp['ray_d'] = b64['ray_d'] * (1.0 + sgn(b64['ray_d']) * 2.0 ** -22)
        bp['ray_o'] = b64['ray_o'] + sgn(b64['ray_o']) * 2.0 ** -23
        bp['near'] = b64['near'] * (1.0 + sgn(b64['near']) * 2.0 ** -23)
        bp['far'] = b64['far'] * (1.0 + sgn(b64['far']) * 2.0 ** -23)
        with torch.no_grad():
            r = O.render(model64, bp, n_samples=n_samples, chunk=chunk)
        noise = torch.maximum(noise, (r['rgb_map'][0] - exact).abs().max(1)[0])
    return noise


def part_field_noise(O, model64, pid, tpts, tdirs, latent_index, ref32=None, trials=4, seed=0, column=None):
    """How ill-conditioned is the field value of part `pid` at the canonical points `tpts` (n,3) / directions `tdirs` (n,3)?  The
    noise scale per point = the largest move of the FLOAT64 value [rgb, occ] (or of output `column` alone: 3 = the occupancy, which
    does not depend on the direction) under `trials` independent fp32-ulp-sized random perturbations of the inputs (what any fp32
    evaluation upstream does to them), together with the deviation of an fp32 evaluation `ref32` (n,4) when given.  Inside the part's
    box this is ~1e-7; outside it the encoder extrapolates (part_base_embedder.py:117-118) with trilinear weights that grow with the
    distance and cancel, and the scale grows with them.  -> (exact (n,4) float64, noise (n,))."""
    sd = model64.sd
    x, d = tpts.double(), tdirs.double()
    ev = lambda xx, dd: O.part_field(xx, dd, sd, pid, model64.pspec[pid], model64.n_occ[pid], model64.n_rgb[pid], latent_index, model64.n_freq)
    sel = (lambda v: v.abs().max(1)[0]) if column is None else (lambda v: v[:, column].abs())
    with torch.no_grad():
        exact = ev(x, d)
        noise = torch.zeros(x.shape[0], dtype=torch.float64)
        if ref32 is not None:
            noise = torch.maximum(noise, sel(ref32.double() - exact))
        g = torch.Generator().manual_seed(2000 + seed)
        sgn = lambda t: (torch.randint(0, 2, t.shape, generator=g).double() * 2.0 - 1.0)
        # one ulp of the coordinate itself and of the box-relative coordinate (|x| + 1 covers the normalisation by the bounds)
        for _ in range(trials):
            r = ev(x + sgn(x) * (x.abs() + 1.0) * 2.0 ** -23, d * (1.0 + sgn(d) * 2.0 ** -22))
            noise = torch.maximum(noise, sel(r - exact))
    return exact, noise
