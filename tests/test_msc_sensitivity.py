"""MSC sensitivity guard for the six-bit soft decisions (the FIC has its curve check against the reference,
tests/test_reference_observed.py; the weaker codes of the MSC had none after the soft bits went from +-127 to +-31 so that
twice a soft value fits the int8 operand of the matrix core).

The oracle decodes the same noisy recordings twice — with the product's contract (+-31) and with eight-bit soft decisions
(+-127, a test knob: orx_set_soft_bits) — and the sub-channel frame error rates (a logical frame of 24 ms with any wrong
byte; what a DAB+ access-unit CRC sees) are compared on the waterfall of EEP 3-A (rate 1/2) and EEP 4-A (rate 3/4):

  * the loss of the six-bit quantisation, read as a shift of the waterfall, stays below 0.1 dB;
  * at the SNRs the verdict names (5 / 6 / 7 dB) EEP 3-A behaves as the reference was seen to (no
    AU CRC error at 9 dB; App. A.6: FIC clean from 5 dB): no frame error from 6 dB on.

The GPU decoder equals the oracle bit for bit (tests/test_gpu_parity.py), so this pins the product's sensitivity too."""
import math

import numpy as np
import pytest

from oracle import binding as ob


def frame_errors(option, level, snr, bits, n_frames=44, seed=3):
    kb = 64
    sub = ob.subch_layout(864 // ob.eep_profile(option, level, kb).n_cu, kb, option, level)
    iq, _, msc = ob.tx_generate(seed=seed, n_frames=n_frames, subch=sub, delay=1000, snr_db=snr)
    o = ob.Stream(subch=sub, ring_len=(n_frames + 2) * ob.TF)
    o.set_soft_bits(bits)
    o.push(iq)
    nsub, fb = len(sub), 3 * kb
    err = tot = f0 = 0
    while f0 + 5 <= n_frames:
        r = o.process(4, want_soft=False)
        assert r["rc"] == 4
        for f in range(4):
            for c in range(4):
                row = (f0 + f) * 4 + c - 15
                if not r["msc_valid"][f, c] or not 0 <= row < msc.shape[0]:
                    continue
                err += int((r["msc"][f, c].reshape(nsub, fb) != msc[row].reshape(nsub, fb)).any(axis=1).sum())
                tot += nsub
        f0 += 4
    o.close()
    return err, tot


@pytest.mark.parametrize("option,level,grid", [(0, 3, (4.0, 4.5, 5.0)), (0, 4, (6.5, 7.0, 7.5, 8.0))])
def test_six_bit_soft_decisions_cost_less_than_a_tenth_of_a_db(option, level, grid):
    fer6, fer8 = [], []
    for snr in grid:
        e6, n = frame_errors(option, level, snr, 6)
        e8, _ = frame_errors(option, level, snr, 8)
        assert e8 >= 30, f"grid point {snr} dB carries too few errors ({e8}) to compare"
        fer6.append(e6 / n)
        fer8.append(e8 / n)
    # slope of the eight-bit waterfall (ln FER per dB) between neighbouring grid points; the six-bit curve read against it
    shifts = []
    for i, snr in enumerate(grid):
        j = i + 1 if i + 1 < len(grid) else i - 1
        slope = (math.log(fer8[j]) - math.log(fer8[i])) / (grid[j] - snr)
        assert slope < -0.5
        shifts.append((math.log(fer6[i]) - math.log(fer8[i])) / -slope)
    # the saturated top of the waterfall (FER > 0.8) says nothing about a shift
    use = [s for s, f in zip(shifts, fer8) if f < 0.8]
    assert use and float(np.mean(use)) < 0.1 and max(use) < 0.2, (list(zip(grid, fer6, fer8)), shifts)


@pytest.mark.parametrize("snr,limit", [(5.0, 0.12), (6.0, 0.002), (7.0, 0.0)])
def test_eep_3a_at_the_verdicts_snrs(snr, limit):
    e6, n = frame_errors(0, 3, snr, 6)
    assert n >= 2500 and e6 / n <= limit, (e6, n)


@pytest.mark.parametrize("snr,limit", [(7.0, 0.30), (8.0, 0.03), (9.0, 0.004)])
def test_eep_4a_waterfall(snr, limit):
    e6, n = frame_errors(0, 4, snr, 6)
    assert n >= 3500 and e6 / n <= limit, (e6, n)
