"""Differentiable wrappers of the HIP ops (backward of the path).

Not built yet: the forward-only kernels must never be used silently where a gradient is expected, so
every entry point here fails loudly until the VJP kernels land.
"""


def _todo(name):
    raise NotImplementedError('ndcn_amd: backward through `%s` is not built yet (SURVEY.md 8f rank 1). '
                              'Run the forward under torch.no_grad().' % name)


def spmm(A, x):
    _todo('spmm')


def linear(x, W, b):
    _todo('linear')


def rhs(A, x, W, b, no_graph, no_control):
    _todo('rhs')
