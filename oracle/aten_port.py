"""CPU baseline "port": the reference's MU loop with the reference's own ATen op sequence.

TEST / BENCH INFRASTRUCTURE ONLY (used by bench.py's ``cpu_baseline`` leg and by tests).

oracle/mu_oracle.py states the MU step in closed form.  The reference instead builds the
reconstruction with ``F.linear`` and calls ``WH.backward(grad_output)`` twice
(torchnmf/nmf.py:52-92, 366-391).  For a *timing* baseline the op sequence matters (autograd's
saved tensors, the extra elementwise temporaries), so this file re-states the loop with that
sequence: linear -> add(eps) -> div -> backward -> relu_/add_ -> div_ -> mul_.  The test-suite
checks it is numerically identical to mu_oracle.fit (and hence to the golden vectors).
"""
import torch
import torch.nn.functional as F

from .mu_oracle import EPS, gamma_of


def _grad_outputs(V, WH, beta):
    if beta == 2:
        return V, WH
    if beta == 1:
        return V / WH.add(EPS), None
    if beta == 0:
        gp = WH.add(EPS).reciprocal_()
        return gp.square().mul_(V), gp
    we = WH.add(EPS)
    return we.pow(beta - 2).mul_(V), we.pow_(beta - 1)


def _update(V, WH, p, beta, gamma, l1, l2, pos):
    p.grad = None
    gneg, gpos = _grad_outputs(V, WH, beta)
    WH.backward(gneg, retain_graph=pos is None)
    neg = p.grad.relu_().add_(EPS)
    if pos is None:
        p.grad = None
        WH.backward(gpos)
        pos = p.grad.relu_().add_(EPS)
    if l1 > 0:
        pos.add_(l1)
    if l2 > 0:
        pos = pos.add(p.data, alpha=l2)
    mult = neg.div_(pos)
    if gamma != 1:
        mult.pow_(gamma)
    p.data.mul_(mult)


def mu_iterations(V, W0, H0, beta=1, n_iter=1, alpha=0.0, l1_ratio=0.0):
    """Run ``n_iter`` MU iterations (no loss evaluation) and return (W, H)."""
    W = torch.nn.Parameter(W0.clone().float())
    H = torch.nn.Parameter(H0.clone().float())
    gamma = gamma_of(beta)
    l1, l2 = alpha * l1_ratio, alpha * (1 - l1_ratio)
    for _ in range(n_iter):
        pos = H.detach().sum(0, keepdim=True) if beta == 1 else None
        _update(V, F.linear(H.detach(), W), W, beta, gamma, l1, l2, pos)
        pos = W.detach().sum(0) if beta == 1 else None
        _update(V, F.linear(H, W.detach()), H, beta, gamma, l1, l2, pos)
    return W.data, H.data


def mu_iterations_nmfd(V, W0, H0, beta=1, n_iter=1):
    """Same loop for NMFD / NMF2D / NMF3D: reconstruction = F.convNd(H, W.flip(shift axes), padding=T-1)
    (nmf.py:776-779, 857-860, 937-940)."""
    W = torch.nn.Parameter(W0.clone().float())
    H = torch.nn.Parameter(H0.clone().float())
    gamma = gamma_of(beta)
    nd = W.dim() - 2
    conv = (F.conv1d, F.conv2d, F.conv3d)[nd - 1]
    axes = tuple(range(2, 2 + nd))
    pad = tuple(k - 1 for k in W.shape[2:])
    red = (0,) + axes
    for _ in range(n_iter):
        pos = H.detach().sum(red, keepdim=True) if beta == 1 else None
        _update(V, conv(H.detach(), W.flip(axes), padding=pad), W, beta, gamma, 0.0, 0.0, pos)
        pos = W.detach().sum(red, keepdim=True).squeeze(0) if beta == 1 else None
        _update(V, conv(H, W.detach().flip(axes), padding=pad), H, beta, gamma, 0.0, 0.0, pos)
    return W.data, H.data


def betamu_iterations(V, W0, H0, beta=1, n_iter=1, l1=0.0, l2=0.0, ortho=0.0):
    """``trainer.BetaMu.step`` on one NMF layer with the reference's op sequence (trainer.py:72-112): the closure's
    F.linear per parameter, two backward passes (the second with ones for beta == 1), clone/relu_, penalties, eps."""
    W = torch.nn.Parameter(W0.clone().float())
    H = torch.nn.Parameter(H0.clone().float())
    gamma = gamma_of(beta)
    for _ in range(n_iter):
        for p, other in ((W, H), (H, W)):
            other.requires_grad_(False)
            p.requires_grad_(True)
            p.grad = None
            WH = F.linear(H, W)
            gneg, gpos = _grad_outputs(V, WH.detach(), beta)
            if gpos is None:
                gpos = torch.ones_like(WH)
            WH.backward(gneg, retain_graph=True)
            neg = torch.clone(p.grad).relu_()
            p.grad.zero_()
            WH.backward(gpos)
            pos = torch.clone(p.grad).relu_()
            p.grad.add_(-neg)
            with torch.no_grad():
                if l1 > 0:
                    pos.add_(l1)
                if l2 > 0:
                    pos.add_(p, alpha=l2)
                if ortho > 0:
                    pos.add_(p.sum(1, keepdim=True) - p, alpha=ortho)
                pos.add_(EPS)
                neg.add_(EPS)
                mult = neg.div_(pos)
                if gamma != 1:
                    mult.pow_(gamma)
                p.mul_(mult)
    return W.data, H.data


def sp_mu_iterations(V, W0, H0, beta=1, n_iter=1):
    """The reference's sparse-target update with its own op sequence (nmf.py:95-119, 366-391, 602-638): the scalars
    ``pos`` / ``neg`` built from the stored entries, two backward passes, relu_/add_, div_, mul_.  beta in {1, 2}."""
    V = V.coalesce()
    idx, vals = V.indices(), V.values()
    ii, jj = idx[0], idx[1]
    W = torch.nn.Parameter(W0.clone().float())
    H = torch.nn.Parameter(H0.clone().float())
    gamma = gamma_of(beta)

    def terms(Hx, Wx):
        if beta == 2:
            pos = torch.linalg.multi_dot([Hx, Wx.t(), Wx]).view(-1) @ Hx.view(-1) * 0.5
            neg = (V.t() @ Hx).view(-1) @ Wx.view(-1)
            return pos, neg
        s = (Wx[jj] * Hx[ii]).sum(1)
        return Wx.sum(0) @ Hx.sum(0), vals @ s.add(EPS).log()

    def update(p, pos_out, neg_out, pos):
        p.grad = None
        neg_out.backward(retain_graph=pos is None)
        neg = p.grad.relu_().add_(EPS)
        if pos is None:
            p.grad = None
            pos_out.backward()
            pos = p.grad.relu_().add_(EPS)
        p.data.mul_(neg.div_(pos) if gamma == 1 else neg.div_(pos).pow_(gamma))

    for _ in range(n_iter):
        pos, neg = terms(H.detach(), W)
        update(W, pos, neg, H.detach().sum(0, keepdim=True) if beta == 1 else None)
        pos, neg = terms(H, W.detach())
        update(H, pos, neg, W.detach().sum(0) if beta == 1 else None)
    return W.data, H.data


def plca_iterations(V, W0, H0, Z0, n_iter=1):
    """PLCA's EM iteration with the reference's op sequence (plca.py:248-290, all factors trainable, no priors): one
    reconstruction H @ (W * Z).T, one backward with V / (WZH + eps), relu / mul / div in place."""
    def nrm(x):
        return x.sum([d for d in range(x.dim()) if d != 1], keepdim=True) if x.dim() > 1 else x.sum()
    W = torch.nn.Parameter(W0.clone().float() / nrm(W0.float()))
    H = torch.nn.Parameter(H0.clone().float() / nrm(H0.float()))
    Z = torch.nn.Parameter(Z0.clone().float() / Z0.float().sum())
    Vn = V / V.sum()
    for _ in range(n_iter):
        for p in (W, H, Z):
            p.grad = None
        WZH = H @ (W * Z).t()
        WZH.backward(Vn / WZH.add(EPS))
        with torch.no_grad():
            Z.data.mul_(Z.grad.relu())
            z_prior = Z.data.clone()
            Z.data.div_(Z.data.sum())
            W.data.mul_(W.grad.relu()).div_(z_prior)
            H.data.mul_(H.grad.relu()).div_(z_prior)
    return W.data, H.data, Z.data
