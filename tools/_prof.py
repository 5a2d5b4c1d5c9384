"""HIP-event breakdown of one call by kernel family (ndcn_prof_*: events recorded on the launch stream around every library
launch, with the launch's algorithmic bytes / flops) - shared by tools/bench_train.py, tools/bench_adjoint.py."""
HBM_PEAK_GBS = 8000.0


def breakdown(fn):
    """runs fn() once under the profiler -> ({family: {launches, ms_total, avg_ms, alg_GBps, frac_of_hbm_peak, TFLOPs}}, total ms)"""
    import torch
    from ndcn_amd import _lib
    lib = _lib.load()
    nk = lib.ndcn_prof_kinds()
    buf = (_lib.ctypes.c_double * (4 * nk))()
    lib.ndcn_prof_enable(1)
    lib.ndcn_prof_read(buf, nk)
    fn()
    torch.cuda.synchronize()
    lib.ndcn_prof_enable(0)
    lib.ndcn_prof_read(buf, nk)
    out, tot = {}, 0.0
    for i, kname in enumerate(_lib.PROF_KINDS):
        cnt, ms, byt, fl = buf[4 * i:4 * i + 4]
        if cnt:
            out[kname] = {'launches': int(cnt), 'ms_total': round(ms, 3), 'avg_ms': round(ms / cnt, 4),
                          'alg_GBps': round(byt / ms / 1e6, 1), 'frac_of_hbm_peak': round(byt / ms / 1e6 / HBM_PEAK_GBS, 4),
                          'TFLOPs': round(fl / ms / 1e9, 2)}
            tot += ms
    return out, tot


def roofline_of(bd, tot, name=None):
    """the bench-line `roofline` object for one family (default: the one with the largest share of the kernel time)"""
    if not bd:
        return None
    dom = name or max(bd, key=lambda k: bd[k]['ms_total'])
    b = bd[dom]
    return {'bound': 'hbm', 'kernel': dom, 'achieved': b['alg_GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': b['frac_of_hbm_peak'],
            'avg_ms': b['avg_ms'], 'launches': b['launches'], 'traffic': None, 'share_of_kernel_time': round(b['ms_total'] / tot, 3)}
