"""CPU oracle for the beta-divergence multiplicative-update (MU) hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
module: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` are allowed to.  It is the checker, never the thing that is
shipped or measured as the product.

What it restates (file:line under /root/reference):

* ``torchnmf/constants.py:3``        eps = float32 machine epsilon (2**-23)
* ``torchnmf/metrics.py:6-96``       kl_div / euclidean / is_div / beta_div
* ``torchnmf/nmf.py:52-92``          _double_backward_update (the MU step)
* ``torchnmf/nmf.py:122-131``        beta == 1 closed-form denominators
* ``torchnmf/nmf.py:297-409``        BaseComponent.fit (dense branch only)
* ``torchnmf/nmf.py:691-693``        NMF.reconstruct   (V ~ H W^T)
* ``torchnmf/nmf.py:776-779``        NMFD.reconstruct  (1-D convolutive)

The reference obtains the MU numerator/denominator through two autograd
``backward`` calls on the reconstruction; for a single linear (or conv1d)
layer those gradients are plain contractions, which is what is written out
here ("closed form").  Parity pinning: the reference has no golden vectors
for this path, so this oracle is pinned against the
reference itself, imported in the build container by
``tools/make_golden.py``; the resulting fixtures live in ``tests/golden/``
and ``tests/test_oracle_golden.py`` re-checks the oracle against them.

Everything is float32 torch on CPU, like the reference.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

EPS = float(torch.finfo(torch.float32).eps)  # constants.py:3


# --------------------------------------------------------------------------
# metrics.py:6-96
# --------------------------------------------------------------------------
def beta_div(x: torch.Tensor, y: torch.Tensor, beta: float) -> torch.Tensor:
    """beta-divergence of reconstruction ``x`` from target ``y`` (metrics.py:60-96)."""
    if beta == 2:  # metrics.py:39
        d = x - y
        return (d * d).sum() * 0.5
    if beta == 1:  # metrics.py:22
        yf = y.reshape(-1)
        return yf @ ((y + EPS).log() - (x + EPS).log()).reshape(-1) - y.sum() + x.sum()
    if beta == 0:  # metrics.py:56-57
        ye, xe = y + EPS, x + EPS
        return (ye / xe).sum() - ye.log().sum() + xe.log().sum() - y.numel()
    xe = x.reshape(-1) + EPS  # metrics.py:85
    yf = y.reshape(-1)
    if beta < 0:  # metrics.py:87-88
        yf = yf + EPS
    bm = beta - 1
    t1 = yf.pow(beta).sum()
    t2 = xe.pow(beta).sum()
    t3 = yf @ xe.pow(bm)
    return (t1 + bm * t2 - beta * t3) / (beta * bm)


def fit_loss(x: torch.Tensor, y: torch.Tensor, beta: float) -> float:
    """The scalar ``fit`` tracks: sqrt(2 * beta_div) (nmf.py:362, 402)."""
    return float((beta_div(x, y, beta) * 2).sqrt())


def gamma_of(beta: float) -> float:
    """MU exponent (nmf.py:341-346)."""
    if beta < 1:
        return 1.0 / (2.0 - beta)
    if beta > 2:
        return 1.0 / (beta - 1.0)
    return 1.0


def mu_terms(V: torch.Tensor, S: torch.Tensor, beta: float):
    """(output_neg, output_pos) of nmf.py:61-74.  ``output_pos`` is None for beta == 1."""
    if beta == 2:
        return V, S  # no eps (nmf.py:62-63)
    if beta == 1:
        return V / (S + EPS), None
    if beta == 0:
        gp = (S + EPS).reciprocal()
        return gp.square() * V, gp
    Se = S + EPS
    return Se.pow(beta - 2) * V, Se.pow(beta - 1)


def _apply(theta: torch.Tensor, neg: torch.Tensor, pos: torch.Tensor, pos_is_closed_form: bool,
           gamma: float, l1: float, l2: float) -> torch.Tensor:
    """nmf.py:78-92: relu/eps, regularisers, multiplier, in-place multiply."""
    neg = neg.relu() + EPS  # nmf.py:78
    if not pos_is_closed_form:
        pos = pos.relu() + EPS  # nmf.py:83 (skipped for the beta == 1 closed form)
    if l1 > 0:
        pos = pos + l1  # nmf.py:85-86
    if l2 > 0:
        pos = pos + l2 * theta  # nmf.py:87-88
    mult = neg / pos
    if gamma != 1:
        mult = mult.pow(gamma)
    return theta * mult


# --------------------------------------------------------------------------
# dense NMF:  V (N,C) ~ H (N,R) @ W (C,R)^T      (nmf.py:659-662, 691-693)
# --------------------------------------------------------------------------
def nmf_reconstruct(H: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    return H @ W.t()


def nmf_w_step(V, W, H, beta, gamma, l1=0.0, l2=0.0):
    """W half-step (nmf.py:367-378).  grad_W of <S, G> is G^T H."""
    S = nmf_reconstruct(H, W)
    gn, gp = mu_terms(V, S, beta)
    neg = gn.t() @ H
    if gp is None:
        pos = H.sum(0, keepdim=True)  # nmf.py:122-125
        return _apply(W, neg, pos, True, gamma, l1, l2)
    return _apply(W, neg, gp.t() @ H, False, gamma, l1, l2)


def nmf_h_step(V, W, H, beta, gamma, l1=0.0, l2=0.0):
    """H half-step with the already updated W (nmf.py:380-391).  grad_H is G W."""
    S = nmf_reconstruct(H, W)
    gn, gp = mu_terms(V, S, beta)
    neg = gn @ W
    if gp is None:
        pos = W.sum(0)  # nmf.py:128-131 -> (R,)
        return _apply(H, neg, pos, True, gamma, l1, l2)
    return _apply(H, neg, gp @ W, False, gamma, l1, l2)


# --------------------------------------------------------------------------
# trainer.BetaMu on one NMF layer (trainer.py:35-121)
# --------------------------------------------------------------------------
def betamu_terms(V: torch.Tensor, S: torch.Tensor, beta: float):
    """(output_neg, output_pos) of trainer.py:75-91: as nmf.py:61-74 except that beta == 1 back-propagates ones."""
    gn, gp = mu_terms(V, S, beta)
    return gn, (torch.ones_like(S) if gp is None else gp)


def betamu_update(theta, neg_raw, pos_raw, gamma, l1=0.0, l2=0.0, ortho=0.0):
    """trainer.py:93-112 for one parameter.  Returns (new theta, p.grad)."""
    neg = neg_raw.relu()                      # trainer.py:94
    pos = pos_raw.relu()                      # trainer.py:97
    grad = pos - neg                          # trainer.py:98
    if l1 > 0:
        pos = pos + l1                        # trainer.py:100-101
    if l2 > 0:
        pos = pos + l2 * theta                # trainer.py:102-103
    if ortho > 0:
        pos = pos + ortho * (theta.sum(1, keepdim=True) - theta)   # trainer.py:105-106
    pos = pos + EPS                           # trainer.py:108
    neg = neg + EPS                           # trainer.py:109
    mult = neg / pos
    if gamma != 1:
        mult = mult.pow(gamma)
    return theta * mult, grad


def betamu_step(V, W, H, beta, l1=0.0, l2=0.0, ortho=0.0, params=('W', 'H')):
    """One ``BetaMu.step`` over ``params`` (in that order; the closure is re-evaluated per parameter, trainer.py:72).
    Returns (W, H, {name: grad})."""
    gamma = gamma_of(beta)
    grads = {}
    for name in params:
        S = nmf_reconstruct(H, W)
        gn, gp = betamu_terms(V, S, beta)
        if name == 'W':
            W, grads['W'] = betamu_update(W, gn.t() @ H, gp.t() @ H, gamma, l1, l2, ortho)
        else:
            H, grads['H'] = betamu_update(H, gn @ W, gp @ W, gamma, l1, l2, ortho)
    return W, H, grads


# --------------------------------------------------------------------------
# NMFD:  V (B,C,L) ~ sum_t W[:,:,t] H[:,:,l-t]    (nmf.py:706-713, 776-779)
#        W (C,R,T), H (B,R,L-T+1)
# --------------------------------------------------------------------------
def nmfd_reconstruct(H: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    B, R, Lh = H.shape
    C, _, T = W.shape
    out = torch.zeros(B, C, Lh + T - 1, dtype=H.dtype)
    for t in range(T):
        # out[b, c, j + t] += sum_r W[c, r, t] H[b, r, j]
        out[:, :, t:t + Lh] += torch.einsum('cr,brj->bcj', W[:, :, t], H)
    return out


def _nmfd_grad_w(G: torch.Tensor, H: torch.Tensor, T: int) -> torch.Tensor:
    # W.grad[c,r,t] = sum_b sum_j G[b,c,j+t] H[b,r,j]            
    Lh = H.shape[2]
    cols = [torch.einsum('bcj,brj->cr', G[:, :, t:t + Lh], H) for t in range(T)]
    return torch.stack(cols, dim=2)


def _nmfd_grad_h(G: torch.Tensor, W: torch.Tensor, Lh: int) -> torch.Tensor:
    # H.grad[b,r,j] = sum_c sum_t W[c,r,t] G[b,c,j+t]
    T = W.shape[2]
    out = torch.zeros(G.shape[0], W.shape[1], Lh, dtype=G.dtype)
    for t in range(T):
        out += torch.einsum('cr,bcj->brj', W[:, :, t], G[:, :, t:t + Lh])
    return out


def nmfd_w_step(V, W, H, beta, gamma, l1=0.0, l2=0.0):
    S = nmfd_reconstruct(H, W)
    gn, gp = mu_terms(V, S, beta)
    T = W.shape[2]
    neg = _nmfd_grad_w(gn, H, T)
    if gp is None:
        pos = H.sum((0, 2), keepdim=True)  # (1,R,1)  nmf.py:122-125
        return _apply(W, neg, pos, True, gamma, l1, l2)
    return _apply(W, neg, _nmfd_grad_w(gp, H, T), False, gamma, l1, l2)


def nmfd_h_step(V, W, H, beta, gamma, l1=0.0, l2=0.0):
    S = nmfd_reconstruct(H, W)
    gn, gp = mu_terms(V, S, beta)
    Lh = H.shape[2]
    neg = _nmfd_grad_h(gn, W, Lh)
    if gp is None:
        pos = W.sum((0, 2), keepdim=True).squeeze(0)  # (R,1)  nmf.py:128-131
        return _apply(H, neg, pos, True, gamma, l1, l2)
    return _apply(H, neg, _nmfd_grad_h(gp, W, Lh), False, gamma, l1, l2)


def betamu_chain_step(V, X0, Ws, beta, l1=0.0, l2=0.0, ortho=0.0, order=None):
    """One ``BetaMu.step`` over a CHAIN of NMF layers, prediction = X0 @ W1^T @ W2^T ... (nn.Sequential of NMF layers,
    tests/test_trainer.py:10-32).  Parameters are updated in ``order`` (names 'X0', 'W1', 'W2', ...; default W1, X0,
    W2, ... = torch's parameter order), each from a freshly evaluated closure (trainer.py:72).  Gradients through the
    chain in closed form:  X_k = X_{k-1} W_k^T,  G_{k-1} = G_k W_k,  dW_k = G_k^T X_{k-1},  dX0 = G_0."""
    Ws = list(Ws)
    gamma = gamma_of(beta)
    if order is None:
        order = ['W1', 'X0'] + [f'W{k}' for k in range(2, len(Ws) + 1)]
    grads = {}
    for name in order:
        xs = [X0]
        for W in Ws:
            xs.append(xs[-1] @ W.t())
        gn, gp = betamu_terms(V, xs[-1], beta)

        def back(G):
            k = len(Ws)
            while True:
                if name == f'W{k}':
                    return G.t() @ xs[k - 1]
                G = G @ Ws[k - 1]
                k -= 1
                if k == 0:
                    return G
        neg, pos = back(gn), back(gp)
        if name == 'X0':
            X0, grads[name] = betamu_update(X0, neg, pos, gamma, l1, l2, ortho)
        else:
            k = int(name[1:])
            Ws[k - 1], grads[name] = betamu_update(Ws[k - 1], neg, pos, gamma, l1, l2, ortho)
    return X0, Ws, grads


# --------------------------------------------------------------------------
# NMF2D / NMF3D (nmf.py:782-942): the same model with 2 / 3 shift axes.  V (B,C,*L), W (C,R,*T), H (B,R,*(L-T+1)),
#   V[b,c,l] ~ sum_{r,t} W[c,r,t] H[b,r,l-t]      (vector l, t; nmf.py:857-860, 937-940: convNd(H, W.flip, pad=T-1))
# Written as explicit shifted accumulations over the taps (any number of axes), like the NMFD functions above.
# --------------------------------------------------------------------------
def _taps(W):
    import itertools
    return itertools.product(*[range(k) for k in W.shape[2:]])


def _win(t, lh):
    """Index tuple selecting the window [t_d, t_d + lh_d) on every shift axis (after the two leading axes)."""
    return (slice(None), slice(None)) + tuple(slice(td, td + n) for td, n in zip(t, lh))


def convnd_reconstruct(H: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    lh, ks = H.shape[2:], W.shape[2:]
    out = torch.zeros(H.shape[0], W.shape[0], *[n + k - 1 for n, k in zip(lh, ks)], dtype=H.dtype)
    for t in _taps(W):
        out[_win(t, lh)] += torch.einsum('cr,br...->bc...', W[(slice(None), slice(None)) + t], H)
    return out


def _convnd_grad_w(G, H, W):
    lh = H.shape[2:]
    out = torch.zeros_like(W)
    for t in _taps(W):
        out[(slice(None), slice(None)) + t] = torch.einsum('bc...,br...->cr', G[_win(t, lh)], H)
    return out


def _convnd_grad_h(G, W, H):
    lh = H.shape[2:]
    out = torch.zeros_like(H)
    for t in _taps(W):
        out += torch.einsum('cr,bc...->br...', W[(slice(None), slice(None)) + t], G[_win(t, lh)])
    return out


def convnd_w_step(V, W, H, beta, gamma, l1=0.0, l2=0.0):
    gn, gp = mu_terms(V, convnd_reconstruct(H, W), beta)
    neg = _convnd_grad_w(gn, H, W)
    if gp is None:   # nmf.py:122-125: H summed over everything but the rank axis, broadcast over (C, R, *T)
        pos = H.sum([0] + list(range(2, H.dim())), keepdim=True)
        return _apply(W, neg, pos, True, gamma, l1, l2)
    return _apply(W, neg, _convnd_grad_w(gp, H, W), False, gamma, l1, l2)


def convnd_h_step(V, W, H, beta, gamma, l1=0.0, l2=0.0):
    gn, gp = mu_terms(V, convnd_reconstruct(H, W), beta)
    neg = _convnd_grad_h(gn, W, H)
    if gp is None:   # nmf.py:128-131
        pos = W.sum([0] + list(range(2, W.dim())), keepdim=True).squeeze(0)
        return _apply(H, neg, pos, True, gamma, l1, l2)
    return _apply(H, neg, _convnd_grad_h(gp, W, H), False, gamma, l1, l2)


def betamu_conv_step(V, W, H, beta, l1=0.0, l2=0.0, ortho=0.0, params=('W', 'H')):
    """One ``trainer.BetaMu.step`` (trainer.py:35-121) over the parameters of ONE convolutive layer -- NMFD / NMF2D / NMF3D,
    prediction = ``convNd(H, W.flip, padding = T - 1)`` (nmf.py:776-779, 857-860, 937-940).  The two backward passes of
    trainer.py:93-97 through the convolution are the tap-wise contractions of ``_convnd_grad_w`` / ``_convnd_grad_h``; beta == 1
    back-propagates ones (no closed-form shortcut in the trainer), the penalties and the orthogonality term (sum over
    dim 1 = the rank axis of W (C, R, *T) and of H (B, R, *L)) enter before eps.  Returns (W, H, {name: p.grad})."""
    gamma = gamma_of(beta)
    grads = {}
    for name in params:
        gn, gp = betamu_terms(V, convnd_reconstruct(H, W), beta)
        if name == 'W':
            W, grads['W'] = betamu_update(W, _convnd_grad_w(gn, H, W), _convnd_grad_w(gp, H, W), gamma, l1, l2, ortho)
        else:
            H, grads['H'] = betamu_update(H, _convnd_grad_h(gn, W, H), _convnd_grad_h(gp, W, H), gamma, l1, l2, ortho)
    return W, H, grads


# --------------------------------------------------------------------------
# fit driver (nmf.py:297-409, dense branch)
# --------------------------------------------------------------------------
_STEPS = {
    'nmf': (nmf_reconstruct, nmf_w_step, nmf_h_step),
    'nmfd': (nmfd_reconstruct, nmfd_w_step, nmfd_h_step),
    'convnd': (convnd_reconstruct, convnd_w_step, convnd_h_step),
}


def validate_target(V: torch.Tensor, beta: float) -> None:
    """nmf.py:329-336."""
    assert bool(torch.all(V >= 0.)), "Target should be non-negative."
    if float(V.min()) == 0 and beta <= 0:
        raise ValueError("When beta <= 0 and V contains zeros, the training process may diverge. "
                         "Please add small values to V, or use a positive beta value.")


def fit(V: torch.Tensor, W0: torch.Tensor, H0: torch.Tensor, beta: float = 1, tol: float = 1e-4,
        max_iter: int = 200, alpha: float = 0, l1_ratio: float = 0, trainable_W: bool = True,
        trainable_H: bool = True, kind: str = 'nmf',
        snapshots: Optional[List[int]] = None) -> Tuple[torch.Tensor, torch.Tensor, int, List[float], dict]:
    """Returns (W, H, n_iter, losses, snaps).

    ``losses[0]`` is loss_init, then one entry per 10th iteration (nmf.py:393-407).
    ``snaps[k]`` = (W, H) after k iterations for k in ``snapshots``.
    """
    recon, w_step, h_step = _STEPS[kind]
    V = V.float()
    W, H = W0.clone().float(), H0.clone().float()
    validate_target(V, beta)
    gamma = gamma_of(beta)
    l1 = alpha * l1_ratio
    l2 = alpha * (1 - l1_ratio)
    loss_init = fit_loss(recon(H, W), V, beta)
    losses = [loss_init]
    prev = loss_init
    snaps = {}
    n_iter = -1
    for n_iter in range(max_iter):
        if trainable_W:
            W = w_step(V, W, H, beta, gamma, l1, l2)
        if trainable_H:
            H = h_step(V, W, H, beta, gamma, l1, l2)
        if snapshots and (n_iter + 1) in snapshots:
            snaps[n_iter + 1] = (W.clone(), H.clone())
        if n_iter % 10 == 9:
            loss = fit_loss(recon(H, W), V, beta)
            losses.append(loss)
            if (prev - loss) / loss_init < tol:
                break
            prev = loss
    return W, H, n_iter + 1, losses, snaps


# --------------------------------------------------------------------------
# sparse-COO target (nmf.py:351-398, 602-638).  The reference differentiates the scalars
#   beta = 1: pos = W.sum(0) . H.sum(0)            neg = sum_nnz v log(WH + eps)          (nmf.py:624-626)
#   beta = 2: pos = 1/2 <H W^T W, H>               neg = <V^T H, W>                        (nmf.py:616-619)
# whose gradients are the dense numerator / denominator terms restricted to the stored entries (zeros of V add
# nothing to a numerator), so the factor updates equal the dense ones (tests/test_nmf_sparse.py:8-37); only the
# tracked loss  V_norm + pos - neg  (nmf.py:172-181, 357, 397) is a different expression from metrics.beta_div.
# --------------------------------------------------------------------------
def sp_terms(idx, vals, W, H, beta):
    """(pos, neg) of nmf.py:605-638 for stored entries (idx (2, nnz), vals)."""
    ii, jj = idx[0], idx[1]
    if beta == 2:
        pos = ((H @ W.t() @ W).reshape(-1) @ H.reshape(-1)) * 0.5
        VtH = torch.zeros(W.shape[0], H.shape[1]).index_add_(0, jj, vals[:, None] * H[ii])
        return pos, VtH.reshape(-1) @ W.reshape(-1)
    s = (W[jj] * H[ii]).sum(1)
    if beta == 1:
        return W.sum(0) @ H.sum(0), vals @ (s + EPS).log()
    # generic beta (nmf.py:628-636): the positive term runs over EVERY entry of the reconstruction
    pos = (H @ W.t() + EPS).pow(beta).sum() / beta
    return pos, vals @ (s + EPS).pow(beta - 1) / (beta - 1)


def sp_v_norm(vals, beta):
    """nmf.py:172-181."""
    if beta == 2:
        return vals @ vals * 0.5
    if beta == 1:
        return vals @ vals.log() - vals.sum()
    return vals.pow(beta).sum() / beta / (beta - 1)


def sp_fit_loss(idx, vals, W, H, beta) -> float:
    pos, neg = sp_terms(idx, vals, W, H, beta)
    return float(((sp_v_norm(vals, beta) + pos - neg) * 2).sqrt())


def _sp_num(idx, vals, W, H, beta, for_w):
    """Gradient of ``neg`` w.r.t. W (for_w) or H: the dense numerator restricted to the stored entries."""
    ii, jj = idx[0], idx[1]
    if beta == 2:
        g = vals
    else:
        s = (W[jj] * H[ii]).sum(1) + EPS
        g = vals / s if beta == 1 else vals * s.pow(beta - 2)
    if for_w:
        return torch.zeros_like(W).index_add_(0, jj, g[:, None] * H[ii])
    return torch.zeros_like(H).index_add_(0, ii, g[:, None] * W[jj])


def sp_w_step(idx, vals, shape, W, H, beta, gamma, l1=0.0, l2=0.0):
    neg = _sp_num(idx, vals, W, H, beta, True)
    if beta == 1:
        return _apply(W, neg, H.sum(0, keepdim=True), True, gamma, l1, l2)
    if beta == 2:     # grad of pos = W H^T H
        return _apply(W, neg, W @ (H.t() @ H), False, gamma, l1, l2)
    return _apply(W, neg, (H @ W.t() + EPS).pow(beta - 1).t() @ H, False, gamma, l1, l2)   # dense positive term


def sp_h_step(idx, vals, shape, W, H, beta, gamma, l1=0.0, l2=0.0):
    neg = _sp_num(idx, vals, W, H, beta, False)
    if beta == 1:
        return _apply(H, neg, W.sum(0), True, gamma, l1, l2)
    if beta == 2:
        return _apply(H, neg, H @ (W.t() @ W), False, gamma, l1, l2)
    return _apply(H, neg, (H @ W.t() + EPS).pow(beta - 1) @ W, False, gamma, l1, l2)


def sp_fit(idx, vals, shape, W0, H0, beta=1, tol=1e-4, max_iter=200, alpha=0, l1_ratio=0):
    """The fit driver of nmf.py:297-409 on a sparse target.  Returns (W, H, n_iter, losses) like ``fit``."""
    W, H = W0.clone().float(), H0.clone().float()
    gamma, l1, l2 = gamma_of(beta), alpha * l1_ratio, alpha * (1 - l1_ratio)
    loss_init = sp_fit_loss(idx, vals, W, H, beta)
    losses, prev, n_iter = [loss_init], loss_init, -1
    for n_iter in range(max_iter):
        W = sp_w_step(idx, vals, shape, W, H, beta, gamma, l1, l2)
        H = sp_h_step(idx, vals, shape, W, H, beta, gamma, l1, l2)
        if n_iter % 10 == 9:
            loss = sp_fit_loss(idx, vals, W, H, beta)
            losses.append(loss)
            if (prev - loss) / loss_init < tol:
                break
            prev = loss
    return W, H, n_iter + 1, losses


# --------------------------------------------------------------------------
# PLCA (plca.py:311-373) fitted by EM (plca.py:193-304):  V / V.sum() ~ H diag(Z) W^T.
# One reconstruction per iteration feeds all three updates (unlike NMF's alternating half-steps): with
# G = Vn / (H diag(Z) W^T + eps) the reference's WZH.backward(G) leaves
#   W.grad = (G^T H) * Z      H.grad = (G W) * Z      Z.grad[r] = sum_{n,c} G[n,c] H[n,r] W[c,r]
# --------------------------------------------------------------------------
def plca_norm(x: torch.Tensor) -> torch.Tensor:
    """get_norm of plca.py:27-35: sum over everything but axis 1 (vectors: total sum)."""
    if x.dim() > 1:
        return x.sum([d for d in range(x.dim()) if d != 1], keepdim=True)
    return x.sum()


def plca_reconstruct(H, W, Z):
    """PLCA: H @ (W * Z).T (plca.py:371-373).  SIPLCA / SIPLCA2 / SIPLCA3: convNd(H, W.flip * Z) (plca.py:447-449,
    522-525, 602-605) = the NMFD-family reconstruction with W scaled by Z along the rank axis."""
    if W.dim() == 2:
        return H @ (W * Z).t()
    return convnd_reconstruct(H, W * Z.view(1, -1, *([1] * (W.dim() - 2))))


def _plca_products(Vn, W, H, Z):
    """(G^T H, G W) generalised: the gradients of the reconstruction w.r.t. W and H, WITHOUT the factor Z."""
    G = Vn / (plca_reconstruct(H, W, Z) + EPS)
    if W.dim() == 2:
        return G.t() @ H, G @ W
    return _convnd_grad_w(G, H, W), _convnd_grad_h(G, W, H)


def _rank_view(Z, x):
    return Z.view(1, -1, *([1] * (x.dim() - 2))) if x.dim() > 1 else Z


def _plca_prior(x, alpha):
    """plca.py:258-260, 272-275, 286-289 without the final renormalisation: add alpha - 1, clamp below at eps."""
    x = x + (alpha - 1)
    return torch.where(x > EPS, x, torch.full_like(x, EPS))   # F.threshold(x, eps, eps)


def plca_em_step(Vn, W, H, Z, W_alpha=1.0, H_alpha=1.0, Z_alpha=1.0, train=(True, True, True)):
    """One EM iteration (plca.py:248-290).  ``train`` = (W, H, Z) trainable flags."""
    tW, tH, tZ = train
    GtH, GW = _plca_products(Vn, W, H, Z)
    Wg, Hg = GtH * _rank_view(Z, W), GW * _rank_view(Z, H)
    Zg = (W * GtH).sum([d for d in range(W.dim()) if d != 1])
    z_prior = None
    if tZ:
        Z = Z * Zg.relu()
        z_prior = Z.clone()
        if Z_alpha != 1:
            Z = _plca_prior(Z, Z_alpha)
        Z = Z / Z.sum()
    if tW:
        W = W * Wg.relu()
        if z_prior is None:
            div = plca_norm(W)
            z_prior = div.reshape(-1)
        else:
            div = _rank_view(z_prior, W)
        W = W / div
        if W_alpha != 1:
            W = _plca_prior(W, W_alpha)
            W = W / plca_norm(W)
    if tH:
        H = H * Hg.relu()
        div = plca_norm(H) if z_prior is None else _rank_view(z_prior, H)
        H = H / div
        if H_alpha != 1:
            H = _plca_prior(H, H_alpha)
            H = H / plca_norm(H)
    return W, H, Z


def plca_loss(V, W, H, Z, norm) -> float:
    return float((beta_div(plca_reconstruct(H, W, Z) * norm, V, 1) * 2).sqrt())   # plca.py:245-246, 294-295 (kl_div; V = Vn * norm)


def plca_fit(V, W0, H0, Z0, tol=1e-4, max_iter=200, W_alpha=1.0, H_alpha=1.0, Z_alpha=1.0, train=(True, True, True)):
    """Returns (W, H, Z, n_iter, norm, losses) -- n_iter is the LAST iteration index, as plca.py:304 returns it."""
    W, H, Z = W0 / plca_norm(W0), H0 / plca_norm(H0), Z0 / plca_norm(Z0)     # the constructor, plca.py:91-121
    norm = V.sum()
    Vn = V / norm
    loss_init = plca_loss(V, W, H, Z, norm)
    losses, prev, n_iter = [loss_init], loss_init, -1
    for n_iter in range(max_iter):
        W, H, Z = plca_em_step(Vn, W, H, Z, W_alpha, H_alpha, Z_alpha, train)
        if n_iter % 10 == 9:
            loss = plca_loss(V, W, H, Z, norm)
            losses.append(loss)
            if (prev - loss) / loss_init < tol:
                break
            prev = loss
    return W, H, Z, n_iter, float(norm), losses


# --------------------------------------------------------------------------
# column-sharded NMF: simulated on one process.
# Shard g owns V[:, Cg] and W[Cg]; H is replicated.  W half-step is local, the
# H half-step sums per-shard partial numerators/denominators (the all-reduce),
# and relu/eps/regularisers are applied after the sum.
# --------------------------------------------------------------------------
def shard_bounds(C: int, world: int) -> List[Tuple[int, int]]:
    base, rem = divmod(C, world)
    out, s = [], 0
    for g in range(world):
        e = s + base + (1 if g < rem else 0)
        out.append((s, e))
        s = e
    return out


def nmf_h_partials(Vg, Wg, H, beta):
    """Per-shard (numerator, denominator) of the H half-step before relu/eps."""
    S = nmf_reconstruct(H, Wg)
    gn, gp = mu_terms(Vg, S, beta)
    if gp is None:
        return gn @ Wg, Wg.sum(0)
    return gn @ Wg, gp @ Wg


def nmf_fit_sharded(V, W0, H0, world: int, beta=1, n_iter=10, alpha=0, l1_ratio=0):
    V = V.float()
    W, H = W0.clone().float(), H0.clone().float()
    gamma = gamma_of(beta)
    l1, l2 = alpha * l1_ratio, alpha * (1 - l1_ratio)
    bounds = shard_bounds(V.shape[1], world)
    for _ in range(n_iter):
        for (s, e) in bounds:
            W[s:e] = nmf_w_step(V[:, s:e], W[s:e], H, beta, gamma, l1, l2)
        num = torch.zeros_like(H)
        den = None
        for (s, e) in bounds:
            n_g, d_g = nmf_h_partials(V[:, s:e], W[s:e], H, beta)
            num += n_g
            den = d_g if den is None else den + d_g
        H = _apply(H, num, den, beta == 1, gamma, l1, l2)
    return W, H
