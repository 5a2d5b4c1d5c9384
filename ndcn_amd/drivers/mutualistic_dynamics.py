"""python -m ndcn_amd.drivers.mutualistic_dynamics ...  (counterpart of the reference's mutualistic_dynamics.py; see dynamics.py)."""
from .dynamics import main

if __name__ == '__main__':
    main('mutualistic')
