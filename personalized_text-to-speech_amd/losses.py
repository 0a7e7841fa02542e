"""Mirror of the reference's losses.py:7-61 (same names, arguments, results).  The only change:
discriminator_loss returns its per-discriminator terms as tensors, not `.item()` floats (12 host
syncs per step in the reference, losses.py:28-29)."""
import torch


def feature_loss(fmap_r, fmap_g):
    cl = getattr(fmap_g, "cl", None)
    if cl is not None and getattr(fmap_r, "cl", None) is cl and cl[0][0].is_cuda:
        from . import reduce                      # both lists come from one MultiPeriodDiscriminator call: fused kernel path
        return reduce.feature_l1(*cl)
    loss = 0
    for dr, dg in zip(fmap_r, fmap_g):
        for rl, gl in zip(dr, dg):
            loss = loss + torch.mean(torch.abs(rl.float().detach() - gl.float()))
    return loss * 2


def _fused_logits(*lists):
    y8 = getattr(lists[0], "y8", None)
    if y8 is not None and all(getattr(l, "y8", None) is y8 for l in lists) and y8[0].is_cuda and len(y8) <= 8:
        return y8
    return None


def discriminator_loss(disc_real_outputs, disc_generated_outputs):
    y8 = _fused_logits(disc_real_outputs, disc_generated_outputs)
    if y8 is not None:                         # both lists come from one MultiPeriodDiscriminator call: one fused pass
        from . import reduce
        loss, terms = reduce.lsgan(y8, 0)
        return loss, [terms[1 + 2 * d] for d in range(len(y8))], [terms[2 + 2 * d] for d in range(len(y8))]
    loss = 0
    r_losses, g_losses = [], []
    for dr, dg in zip(disc_real_outputs, disc_generated_outputs):
        r_loss = torch.mean((1 - dr.float()) ** 2)
        g_loss = torch.mean(dg.float() ** 2)
        loss = loss + (r_loss + g_loss)
        r_losses.append(r_loss.detach())
        g_losses.append(g_loss.detach())
    return loss, r_losses, g_losses


def generator_loss(disc_outputs):
    y8 = _fused_logits(disc_outputs)
    if y8 is not None:
        from . import reduce
        loss, terms = reduce.lsgan(y8, 1)
        return loss, [terms[2 + 2 * d] for d in range(len(y8))]
    loss = 0
    gen_losses = []
    for dg in disc_outputs:
        l = torch.mean((1 - dg.float()) ** 2)
        gen_losses.append(l)
        loss = loss + l
    return loss, gen_losses


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):
    """z_p, logs_q, m_p, logs_p: [b, h, t_t] (argument order of reference losses.py:46)."""
    z_p, logs_q, m_p, logs_p, z_mask = (t.float() for t in (z_p, logs_q, m_p, logs_p, z_mask))
    kl = logs_p - logs_q - 0.5
    kl = kl + 0.5 * ((z_p - m_p) ** 2) * torch.exp(-2.0 * logs_p)
    if kl.is_cuda:
        from . import reduce
        return reduce.sum_all(kl * z_mask) / reduce.sum_all(z_mask)
    return torch.sum(kl * z_mask) / torch.sum(z_mask)
