"""Host-side diffusion schedules and per-step coefficient tables (tiny, computed once).

These are the numbers the reference keeps in ``GaussianDiffusion`` / ``DDIMSampler``; they are
prepared on the host in the reference's own precision order and uploaded as small device tables
that the step plans index with the on-device step counter.

  * layout: ``get_betas`` (every schedule type the reference can build) + ``GaussianDiffusion.__init__`` and the
    mean / variance parameterisations of ``p_mean_variance``
    (model/networks/diffusion_layout/diffusion_ddpm.py:38-84, 133-162, 220-264)
  * shape : ``make_beta_schedule('linear')``, ``make_ddim_timesteps('uniform')``,
    ``make_ddim_sampling_parameters`` (diffusion_shape/ldm_diffusion_util.py:43-96) and
    ``DDIMSampler.make_schedule`` (samplers/ddim.py:28-57), eta = 0
  * ``timestep_embedding`` (ldm_diffusion_util.py:174-194): a sinusoid table, one row per loop
    iteration.  It is computed on the host because an ulp in the frequency is amplified by t<=999
    (phase error ~6e-5 rad); a table is bit-identical to the reference's CPU value.
"""
import math
import numpy as np
import torch


def timestep_embedding_table(timesteps, dim, max_period=10000):
    t = torch.as_tensor(np.asarray(timesteps), dtype=torch.float32)
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb.contiguous()


def layout_betas(schedule_type, beta_start, beta_end, time_num):
    """``get_betas`` (diffusion_ddpm.py:38-84), float64.  'linear' and the three 'warm*' ramps (a linear ramp over the first 10 / 20 /
    50 % of the steps, beta_end after it).  'cosine' never reaches the loop in the reference either: its branch computes the table
    without binding it and the function fails at ``return betas`` (diffusion_ddpm.py:59-80) -- the same exception type is raised here."""
    if schedule_type == 'linear':
        return np.linspace(beta_start, beta_end, time_num).astype(np.float64)
    if schedule_type in ('warm0.1', 'warm0.2', 'warm0.5'):
        betas = beta_end * np.ones(time_num, dtype=np.float64)
        warm = int(time_num * float(schedule_type[4:]))
        betas[:warm] = np.linspace(beta_start, beta_end, warm, dtype=np.float64)
        return betas
    if schedule_type == 'cosine':
        raise UnboundLocalError("schedule_type 'cosine': the reference's get_betas returns an unbound table for it "
                                "(diffusion_ddpm.py:59-84); no model can have been trained with it")
    raise NotImplementedError(schedule_type)


class LayoutSchedule:
    """Per-iteration coefficients of the ancestral DDPM loop, iteration i <-> t = T-1-i: five numbers per step,
    ``x0 = c0 * x - c1 * out;  mean = c2 * x0 + c3 * x;  x' = mean + c4 * noise`` (k_ddpm_update, fp contraction off).

    ``model_mean_type`` (p_mean_variance, diffusion_ddpm.py:239-257): 'eps' -> c0, c1 = sqrt(1 / ac), sqrt(1 / ac - 1);
    'x0' (the network predicts x_0 itself) -> c0, c1 = 0, -1: ``0 * x - (-1 * out)`` is ``out`` bit for bit, so the update kernel is
    the same.  ``model_var_type`` (diffusion_ddpm.py:224-235): 'fixedsmall' -> the clipped posterior log-variance; 'fixedlarge' ->
    log(cat[posterior_variance[1:2], betas[1:]]).  Either way sigma = exp(0.5 * log-variance) and no noise at t == 0."""

    def __init__(self, time_num=1000, beta_start=1e-4, beta_end=0.02, schedule_type='linear', model_mean_type='eps',
                 model_var_type='fixedsmall'):
        if model_mean_type not in ('eps', 'x0'):
            raise NotImplementedError(model_mean_type)          # (as p_mean_variance does at its first call)
        if model_var_type not in ('fixedsmall', 'fixedlarge'):
            raise NotImplementedError(model_var_type)
        self.time_num = time_num
        betas64 = layout_betas(schedule_type, beta_start, beta_end, time_num)
        assert (betas64 > 0).all() and (betas64 <= 1).all()                  # diffusion_ddpm.py:134
        alphas64 = 1.0 - betas64
        ac = torch.from_numpy(np.cumprod(alphas64, axis=0)).float()        # cast to fp32 FIRST
        ac_prev = torch.from_numpy(np.append(1.0, ac[:-1])).float()
        betas = torch.from_numpy(betas64).float()
        alphas = torch.from_numpy(alphas64).float()
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        if model_var_type == 'fixedsmall':
            logvar = torch.log(torch.max(post_var, 1e-20 * torch.ones_like(post_var)))
        else:
            logvar = torch.log(torch.cat([post_var[1:2], betas[1:]]))
        if model_mean_type == 'eps':
            srac = torch.sqrt(1.0 / ac)
            srm1 = torch.sqrt(1.0 / ac - 1)
        else:
            srac = torch.zeros_like(ac)
            srm1 = -torch.ones_like(ac)
        c1 = betas * torch.sqrt(ac_prev) / (1.0 - ac)
        c2 = (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac)
        sigma = torch.exp(0.5 * logvar)
        sigma[0] = 0.0                                                       # no noise when t == 0
        tab = torch.stack([srac, srm1, c1, c2, sigma], dim=1)               # indexed by t
        self.timesteps = np.arange(time_num - 1, -1, -1)                     # iteration order
        self.coef = tab[torch.from_numpy(self.timesteps.copy())].contiguous()   # [T, 5] by iteration


class ShapeSchedule:
    """DDIM coefficients, iteration i <-> index = S-1-i, timestep ts[index].  ``eta`` = 0 (the shipped call, echo2shape.py:484-521):
    four coefficients per step; ``eta`` != 0: sigma_t as make_ddim_sampling_parameters derives it (ldm_diffusion_util.py:85-96: float64
    arithmetic on the fp32 alphas, ``1 - alphas`` and its reciprocal formed in fp32 first, the result cast to fp32 when it is used) and
    sqrt(1 - a_prev - sigma_t^2) in fp32 as p_sample_ddim forms it (samplers/ddim.py:256): five coefficients per step."""

    def __init__(self, ddim_steps=100, timesteps=1000, linear_start=0.00085, linear_end=0.012, eta=0.0):
        betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps,
                                dtype=torch.float64) ** 2).numpy()
        ac = torch.tensor(np.cumprod(1.0 - betas, axis=0), dtype=torch.float32)
        c = timesteps // ddim_steps
        ts = np.asarray(list(range(0, timesteps, c))) + 1
        if ts.max() >= timesteps:
            # same failure the reference hits (IndexError in make_ddim_sampling_parameters)
            raise IndexError('ddim_steps=%d yields timestep %d >= %d' % (ddim_steps, ts.max(), timesteps))
        a = ac[ts]
        prev_list = [ac[0].item()] + ac[ts[:-1]].tolist()
        a_prev = torch.tensor(prev_list, dtype=torch.float32)
        s1m = torch.sqrt(1.0 - a)
        self.eta = float(eta)
        ap64 = np.asarray(prev_list, dtype=np.float64)
        # (the reference evaluates ``ndarray / Tensor``, i.e. Tensor.__rtruediv__ = reciprocal(1 - alphas) in fp32, times the array)
        sig64 = self.eta * np.sqrt((1.0 - a).reciprocal().double().numpy() * (1 - ap64) * (1 - a.double().numpy() / ap64))
        sig = torch.from_numpy(sig64).to(torch.float32)
        self.ddim_sigmas = sig
        cols = [s1m, a.sqrt(), a_prev.sqrt(), (1.0 - a_prev - sig ** 2).sqrt()]
        if self.eta != 0.0:
            cols.append(sig)
        tab = torch.stack(cols, dim=1)
        order = np.arange(len(ts) - 1, -1, -1)
        self.ddim_timesteps = ts
        self.timesteps = ts[order]
        self.coef = tab[torch.from_numpy(order.copy())].contiguous()        # [S, 4 | 5] by iteration
        self.alphas_cumprod = ac
