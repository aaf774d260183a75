"""Host logic of the step recorder (sup3r_amd/captured.py) without a device:
the record key follows every piece of state a graph bakes in, and the cache of
records is bounded (unbounded ``_entries`` under an adaptive
adversarial weight; loss terms / precision missing from the key)."""
import numpy as np

from sup3r_amd import captured


class _Dev:
    options_key = ()
    nranks = 1


class _Net:
    def __init__(self, precision='bf16'):
        self.precision = precision
        self.plan_epoch = 0


class _Compute:
    share_dtrue_allowed = True

    def __init__(self):
        self.dev = _Dev()
        self.gen, self.disc = _Net(), _Net()


class _Opt:
    KIND = 0
    iterations = 5


class _Model:
    def __init__(self):
        self._loss_terms = [('MeanAbsoluteError', 0, 1.0, {})]
        self.optimizer, self.optimizer_disc = _Opt(), _Opt()


class _Batch:
    low_res = np.zeros((2, 3, 3, 1), np.float32)
    high_res = np.zeros((2, 6, 6, 1), np.float32)


def _recorder(monkeypatch):
    rec = captured.StepRecorder(_Compute())
    made = []

    def fake_record(batch, body, nets):
        r = captured._Recorded()
        made.append(r)
        return r
    monkeypatch.setattr(rec, '_record', fake_record)
    monkeypatch.setattr(rec, '_replay', lambda r, batch, nets: ['replayed'])
    monkeypatch.setattr(rec, '_resident', lambda batch: batch)
    return rec, made


def _steps(rec, model, weight, n, opts=None):
    out = []
    for _ in range(n):
        out.append(rec.run(_Batch, (True, True, True, float(weight)),
                           opts or [model.optimizer, model.optimizer_disc],
                           lambda b: ['eager'], model=model))
    return out


def test_records_are_bounded_under_a_moving_adversarial_weight(monkeypatch):
    rec, made = _recorder(monkeypatch)
    m = _Model()
    for i in range(40):                  # update_adversarial_weights per epoch
        got = _steps(rec, m, 1e-3 * (1 + i), 4)
        assert got == [['eager'], ['eager'], ['replayed'], ['replayed']]
    live = [e for e in rec._entries.values() if e['rec'] is not None]
    assert len(made) == 40
    assert len(live) <= rec.MAX_RECORDS
    assert rec.evicted == 40 - len(live)
    assert len(rec._entries) <= 8 * rec.MAX_RECORDS
    # the most recent weights are still recorded (LRU, not FIFO of inserts)
    assert _steps(rec, m, 1e-3 * 40, 1) == [['replayed']]
    _steps(rec, m, 1e-3 * 36, 1)          # touch an older one ...
    for i in range(40, 40 + rec.MAX_RECORDS - 1):
        _steps(rec, m, 1e-3 * (1 + i), 3)
    assert _steps(rec, m, 1e-3 * 36, 1) == [['replayed']]   # ... it survived


def test_key_follows_loss_terms_precision_and_dtrue_sharing(monkeypatch):
    rec, made = _recorder(monkeypatch)
    m = _Model()
    assert _steps(rec, m, 1e-2, 3)[-1] == ['replayed']
    m._loss_terms = [('MeanSquaredError', 1, 1.0, {})]
    assert _steps(rec, m, 1e-2, 3) == [['eager'], ['eager'], ['replayed']]
    m._loss_terms = [('MeanSquaredError', 1, 0.5, {})]
    assert _steps(rec, m, 1e-2, 1) == [['eager']]
    m._loss_terms = [('MeanSquaredError', 1, 1.0, {})]
    assert _steps(rec, m, 1e-2, 1) == [['replayed']]
    rec.compute.gen.precision = 'bf16x3'
    assert _steps(rec, m, 1e-2, 1) == [['eager']]
    rec.compute.gen.precision = 'bf16'
    rec.compute.share_dtrue_allowed = False
    assert _steps(rec, m, 1e-2, 1) == [['eager']]
    rec.compute.share_dtrue_allowed = True
    assert _steps(rec, m, 1e-2, 1) == [['replayed']]


def test_stale_records_go_at_once(monkeypatch):
    rec, made = _recorder(monkeypatch)
    m = _Model()
    _steps(rec, m, 1e-2, 3)
    _steps(rec, m, 2e-2, 3)
    assert sum(e['rec'] is not None for e in rec._entries.values()) == 2
    # update_optimizer replaces the object: the old records can never replay
    m.optimizer = _Opt()
    _steps(rec, m, 1e-2, 1)
    assert sum(e['rec'] is not None for e in rec._entries.values()) == 0
    assert rec.evicted == 2
    _steps(rec, m, 1e-2, 2)
    # clear_plans bumps the plan epoch
    rec.compute.gen.plan_epoch += 1
    _steps(rec, m, 1e-2, 1)
    assert sum(e['rec'] is not None for e in rec._entries.values()) == 0
    assert rec.evicted == 3
