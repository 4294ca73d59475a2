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


def test_vio_front_end_on_the_data_of_a_chain():
    """propagateImuState / initializeImu / updateVel / tryTrapScale (FS/HessianBlocks.cpp:225-429) of the facade on every call the
    oracle's chain makes (NumPy front-end, oracle/imu_frontend.py): sample lists as the hand-over on frame marginalisation leaves
    them, biases and spline states as the optimisation leaves them."""
    import copy
    worst = dict(propagate=0.0, initialize=0.0, update_vel=0.0, trap=0.0)
    calls = dict(propagate=0, initialize=0, update_vel=0, trap=0)

    def rel(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))

    class Both(rolling.OracleChain):
        host = host
        _fe = rolling.DeviceChain._fe
        _shell = rolling.DeviceChain._shell

        def _snap(self):
            return copy.deepcopy((self.imu_state, self.imu_zero, {f: s["vel"].copy() for f, s in self.shells.items()}, dict(self.cal),
                                  np.array(self.scale_queue), self.scale_qi))

        def _restore(self, s):
            self.imu_state, self.imu_zero = copy.deepcopy(s[0]), copy.deepcopy(s[1])
            for f, v in s[2].items():
                self.shells[f]["vel"] = v.copy()
            self.cal = dict(s[3])
            self.scale_queue, self.scale_qi = np.array(s[4]), s[5]

        def _both(self, name, what, *a):
            before = self._snap()
            r_f = getattr(rolling.DeviceChain, name)(self, *a)
            after_f = self._snap()
            self._restore(before)
            r_o = getattr(rolling.OracleChain, name)(self, *a)
            after_o = self._snap()
            calls[what] += 1
            for f in after_o[0]:
                worst[what] = max(worst[what], rel(after_f[0][f], after_o[0][f]), rel(after_f[1][f], after_o[1][f]), rel(after_f[2][f], after_o[2][f]))
            worst[what] = max(worst[what], rel([after_f[3]["scale"], after_f[3]["scale_zero"]], [after_o[3]["scale"], after_o[3]["scale_zero"]]))
            assert after_f[3]["trapped"] == after_o[3]["trapped"] and after_f[5] == after_o[5]
            worst[what] = max(worst[what], rel(after_f[4], after_o[4]))
            assert r_f == r_o or (r_f is None and r_o is None)
            return r_o

        def vio_propagate(self, *a): return self._both("vio_propagate", "propagate", *a)
        def vio_initialize(self, *a): return self._both("vio_initialize", "initialize", *a)
        def vio_update_vel(self, *a): return self._both("vio_update_vel", "update_vel", *a)
        def vio_try_trap(self, *a): return self._both("vio_try_trap", "trap", *a)

    sc = rolling.Scenario(n_frames=24, vio=True)
    ch = Both(sc)
    ch.bootstrap()
    while ch.next_frame < sc.n_frames:
        ch.step()
    print(calls, worst)
    assert calls["propagate"] >= 14 and calls["initialize"] == 1 and calls["update_vel"] >= 18 and calls["trap"] >= 5
    assert max(worst.values()) < 1e-9, worst
