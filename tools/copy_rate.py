"""Device -> page-locked-host copy rate of this box, as the denominator of the `with_d2h` bench rows.

Round 3 timed ONE cold copy (first use of the pinned buffer and of the copy path) and printed fractions of 1.57 and 1.84
"of the copy rate": a denominator that is not the bus.  Here: the copy goes into the same pinned buffer the timed call
writes, is warmed once, repeated, and the BEST rate counts; `copy_rate_fraction` refuses to print a fraction above
1.05 (it re-measures once with the caller's warm buffers and otherwise returns None with the reason)."""
import time


def d2h_copy_rate_gbs(torch, host, dev, reps=4):
    """Best rate over `reps` copies of `dev` (CUDA tensor, >= 2 GiB wanted) into `host[:len(dev)]` (pinned), after one
    warm-up copy."""
    n = dev.shape[0]
    nbytes = dev.numel() * dev.element_size()
    host[:n].copy_(dev, non_blocking=True)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        host[:n].copy_(dev, non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, nbytes / (time.perf_counter() - t0) / 1e9)
    return best


def copy_rate_fraction(torch, achieved_gbs, rate_gbs, host, dev):
    """(rate, fraction or None, note).  A pipelined call cannot beat the bus: a fraction above 1.05 means the rate was
    measured badly, so it is measured again (buffers are warm now) and the larger value is kept."""
    if achieved_gbs / rate_gbs > 1.05:
        rate_gbs = max(rate_gbs, d2h_copy_rate_gbs(torch, host, dev, reps=6))
    frac = achieved_gbs / rate_gbs
    if frac > 1.05:
        return rate_gbs, None, "copy rate measured below the call's own rate twice: not a bound on this box, fraction withheld"
    return rate_gbs, round(frac, 3), "best of >= 4 warm copies of %.1f GiB into the call's own pinned buffer" % (
        dev.numel() * dev.element_size() / 2 ** 30)
