"""torchnmf_amd -- MI355X-native multiplicative-update engine behind torchnmf's ``NMF`` / ``NMFD`` surface.

    from torchnmf_amd.nmf import NMF
    m = NMF(V.shape, rank=128).cuda()
    n_iter = m.fit(V.cuda(), beta=1)

The beta-divergence MU hot path of yoyololicon/pytorch-NMF (dense and sparse-COO ``NMF``, ``NMFD``, ``NMF2D`` / ``NMF3D``,
``trainer.BetaMu`` on layer chains, ``PLCA`` / ``SIPLCA*``) on hand-written HIP kernels; see DESIGN.md for what is out of scope.
"""
name = 'torchnmf_amd'
__version__ = '0.1.0'

from . import constants, metrics, nmf, plca, trainer  # noqa: E402,F401
