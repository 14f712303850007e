"""Backbone / model_sampling stand-ins shared by the golden generator, the oracle
tests and the GPU parity tests.  They follow the stub shape the reference's own
tests use (tests/test_lanpaint_semantic_stop.py:6-17, tests/test_av_schedule.py:110-131):
    model(x, t, model_options=None, seed=None) -> Tensor | (x0, x0_BIG)
    model.inner_model.model_sampling.noise_scaling(sigma, noise, latent_image)
Every stub is written with operators only so the same object works on torch
tensors (reference, product) and numpy arrays (oracle)."""
from __future__ import annotations


def _row(t, x):
    return t.reshape((-1,) + (1,) * (x.ndim - 1))


class VESampling:
    """EPS-style: latent + noise * sigma."""
    lanpaint_noise_scaling_kind = "ve"

    def noise_scaling(self, sigma, noise, latent_image, max_denoise=False):
        return latent_image + noise * sigma


class FlowSampling:
    """CONST-style: sigma * (ns * noise) + (1 - sigma) * latent."""
    lanpaint_noise_scaling_kind = "flow"
    noise_scale = 1.0

    def noise_scaling(self, sigma, noise, latent_image, max_denoise=False):
        return sigma * (self.noise_scale * noise) + (1.0 - sigma) * latent_image


class OpaqueVESampling:
    """Same arithmetic as VESampling but NOT declared: forces the product engine
    through the generic `noise_scaling` callback path."""

    def noise_scaling(self, sigma, noise, latent_image, max_denoise=False):
        return latent_image + noise * sigma


class _Base:
    def __init__(self, flow=False, sampling=None):
        self.inner_model = self
        self.model_sampling = sampling if sampling is not None else (FlowSampling() if flow else VESampling())
        self.calls = 0
        self.last_input = None
        self.last_t = None

    def _note(self, x, t):
        self.calls += 1
        self.last_input = x
        self.last_t = t


class LinearTupleModel(_Base):
    """x -> (0.9 x, 0.8 x): the stub BASELINE.md / SURVEY.md section 8d name."""

    def __call__(self, x, t, model_options=None, seed=None):
        self._note(x, t)
        return 0.9 * x, 0.8 * x


class DenoiserSingleModel(_Base):
    """Single-tensor output depending on t: x / (1 + t^2) (the exact VE denoiser
    for unit-Gaussian data) -- checks that t reaches the model per row."""

    def __call__(self, x, t, model_options=None, seed=None):
        self._note(x, t)
        return x / (1.0 + _row(t, x) ** 2)


class ListOneModel(_Base):
    """Returns a 1-element list (lanpaint.py:39-40 branch)."""

    def __call__(self, x, t, model_options=None, seed=None):
        self._note(x, t)
        return [0.5 * x + 0.1]


class OffsetTupleModel(_Base):
    """(x + a, x - b) with distinct heads."""

    def __call__(self, x, t, model_options=None, seed=None):
        self._note(x, t)
        return x * 0.6 + 0.25, x * 0.4 - 0.125


MODELS = {
    "linear_tuple": LinearTupleModel,
    "denoiser_single": DenoiserSingleModel,
    "list_one": ListOneModel,
    "offset_tuple": OffsetTupleModel,
}
