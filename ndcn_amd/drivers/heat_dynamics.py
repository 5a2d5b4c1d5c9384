"""python -m ndcn_amd.drivers.heat_dynamics ...  (counterpart of the reference's heat_dynamics.py; see dynamics.py)."""
from .dynamics import main

if __name__ == '__main__':
    main('heat')
