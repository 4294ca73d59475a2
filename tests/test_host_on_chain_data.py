"""The facade's host functions that need no device, on the data of a RUNNING chain instead of synthetic inputs: every candidate
selection of activatePointsMT and every IMU-form frame marginalisation of the oracle's rolling visual-inertial chain is handed to
the facade's implementation as well (sosf_activate_select with its bitmap distance transform, sosf_imu_marginalize_frame with its
blocked Schur update).  Priors out of real marginalisations span 1e8 .. 1e-19; candidate sets carry the trace states real traces
leave.  CPU only."""
import numpy as np

from oracle import oracle as orc
from sos_slam_amd import host
from tests import rolling


def test_selection_and_imu_marginalisation_on_the_data_of_a_chain(monkeypatch):
    seen = dict(select=0, optimize=0, marg=0, worst_H=0.0, worst_b=0.0)
    orc_select = orc.activate_select

    def select_both(w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, flagged):
        d_o, D_o = orc_select(w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, flagged)
        d_f, D_f = host.activate_select(w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, flagged)
        assert np.array_equal(d_o, d_f), (seen["select"], int((d_o != d_f).sum()))
        assert np.array_equal(D_o, D_f)
        seen["select"] += 1
        seen["optimize"] += int((d_o == 1).sum())
        return d_o, D_o

    monkeypatch.setattr(rolling.orc, "activate_select", select_both)
    orc_imu = orc.imu
    fac = host.imu()

    class Both:
        def __init__(self):
            self.o = orc_imu()

        def __getattr__(self, name):
            return getattr(self.o, name)

        def marginalize_frame(self, S, cal, recs, idx, delta, pr, dp, HM, bM, marg_weight=0.25):
            Ho, bo = self.o.marginalize_frame(S, cal, recs, idx, delta, pr, dp, HM, bM, marg_weight=marg_weight)
            Hf, bf = fac.marginalize_frame(S, cal, recs, idx, delta, pr, dp, HM, bM, marg_weight=marg_weight)
            seen["marg"] += 1
            seen["worst_H"] = max(seen["worst_H"], np.abs(Ho - Hf).max() / np.abs(Ho).max())
            seen["worst_b"] = max(seen["worst_b"], np.abs(bo - bf).max() / np.abs(bo).max())
            assert np.array_equal(Hf, Hf.T)
            return Ho, bo

    monkeypatch.setattr(rolling.orc, "imu", Both)
    sc = rolling.Scenario(n_frames=20, vio=True)
    ch = rolling.OracleChain(sc)
    ch.bootstrap()
    while ch.next_frame < sc.n_frames:
        ch.step()
    print(seen)
    assert seen["select"] >= 14 and seen["optimize"] > 500 and seen["marg"] >= 6
    assert seen["worst_H"] < 1e-9 and seen["worst_b"] < 1e-9
