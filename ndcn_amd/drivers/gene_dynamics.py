"""python -m ndcn_amd.drivers.gene_dynamics ...  (counterpart of the reference's gene_dynamics.py; see dynamics.py)."""
from .dynamics import main

if __name__ == '__main__':
    main('gene')
