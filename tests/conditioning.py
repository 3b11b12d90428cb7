"""Test infrastructure: how ill-conditioned is a pixel of the reference's arithmetic?

Band / far pairs land far outside a part's box and are EXTRAPOLATED by the encoder with trilinear weights of 1e3..1e9 (DESIGN.md
§3): a one-ulp difference anywhere upstream moves such a pixel by 1e-4 and more — in the reference's own fp32 run as much as in the
kernels.  One fp32 run of the oracle is a single noisy sample of that (and changes with the host's thread count); so the noise scale of
a pixel is estimated as the largest move of the FLOAT64 result under several (default 4) independent fp32-ulp-sized random perturbations of the
rays, together with the deviation of the oracle's fp32 run.  Checker only."""
import torch


def pixel_noise(O, model64, b64, exact, n_samples, chunk=64, ref32=None, trials=4, seed=0):
    """exact: (n,3) float64 rgb_map of the unperturbed float64 run -> (n,) noise scale per pixel."""
    g = torch.Generator().manual_seed(1000 + seed)
    noise = torch.zeros(exact.shape[0], dtype=torch.float64)
    if ref32 is not None:
        noise = torch.maximum(noise, (ref32.double() - exact).abs().max(1)[0])
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
