"""odeint when a gradient is required (backprop through the solver, as the reference drivers train:
heat_dynamics.py:333).  SURVEY.md 8f rank 1 - not built yet; fails loudly instead of returning a
forward-only result that autograd would treat as a constant."""


def odeint_with_grad(func, y0, t, rtol, atol, method, options, autonomous=False):
    raise NotImplementedError('ndcn_amd.odeint: backward through the HIP solver is not built yet '
                              '(SURVEY.md 8f rank 1). Run under torch.no_grad() / with parameters frozen.')
