"""DDPMScheduler, 1 inference step (test infrastructure).

Restates diffusers 0.25.1 ``scheduling_ddpm.py`` as configured by the
reference's ``make_1step_sched`` (src/model.py:7-11) from the sd-turbo
scheduler config: scaled_linear betas 0.00085..0.012, 1000 train steps,
epsilon prediction, clip_sample False, trailing spacing, fixed_small variance
(SURVEY.md A.6).  ``set_timesteps(1)`` -> [999]; step(eps, 999, x):
prev_t = -1 -> alpha_prod_prev = 1 -> prev_sample = pred_x0 + 1e-10 * noise.
"""
import torch


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def one_step_constants(t=999):
    """(sqrt(abar_t), sqrt(1-abar_t)) as python floats from the fp32 table."""
    ac = alphas_cumprod()
    a = ac[t]
    return float(a ** 0.5), float((1 - a) ** 0.5)


def ddpm_step(model_output, sample, t=999, variance_noise=None):
    """DDPMScheduler.step for the single inference timestep.

    pred_x0 = (x - sqrt(1-abar) eps) / sqrt(abar); coefficients for t-1 = -1 are
    (1, 0); variance = clamp(0, min=1e-20) -> std 1e-10 (numerically dead,
    A.9 quirk 2; applied only when ``variance_noise`` is given).
    """
    ac = alphas_cumprod()
    a_t = ac[t]
    a_prev = torch.tensor(1.0)
    b_t = 1 - a_t
    b_prev = 1 - a_prev
    cur_alpha = a_t / a_prev
    cur_beta = 1 - cur_alpha
    pred_x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
    c_x0 = (a_prev ** 0.5 * cur_beta) / b_t
    c_xt = cur_alpha ** 0.5 * b_prev / b_t
    prev = c_x0 * pred_x0 + c_xt * sample
    if variance_noise is not None:
        var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_beta, min=1e-20)
        prev = prev + var ** 0.5 * variance_noise
    return prev
