"""Training bookkeeping of the MI355X Sup3rGan: running loss windows and the
per-epoch history table.

The reference keeps two pandas frames and appends one row per mini-batch with
``record.loc[...]`` (sup3r/models/abstract.py:589-622); at MI355X step times
(6 ms per C1 batch) that host work is a visible share of the step.  Here the
per-batch record is a fixed-size numpy ring per loss key — appending is one
array store, the running mean one ``nanmean`` over <= n_batches values — and
pandas is touched once per epoch, when the history row is written.  What is
preserved is the *observable* behaviour: the running means cover the last
``n_batches`` mini-batches, a key that was not computed in a mini-batch is
carried at its previous mean, and the history schema (``train_*`` / ``val_*``
columns, ``gen_train_frac`` ...) is the reference's.
"""
import numpy as np
import pandas as pd


class LossWindow:
    """Sliding window over the last ``size`` mini-batches of named scalars."""

    def __init__(self, prefix, size=1):
        self.prefix = prefix
        self._size = max(1, int(size))
        self._rows = {}        # key -> float64 ring (NaN = not recorded)
        self._n = 0            # rows pushed so far

    # ---- window geometry
    def resize(self, size):
        size = max(1, int(size))
        if size == self._size:
            return
        keep = min(self._n, self._size, size)
        new = {}
        for k in self._rows:
            vals = self._ordered(k)[-keep:] if keep else []
            ring = np.full(size, np.nan)
            ring[:len(vals)] = vals
            new[k] = ring
        self._rows, self._size, self._n = new, size, keep

    def _ordered(self, key):
        ring = self._rows[key]
        n = min(self._n, self._size)
        if self._n <= self._size:
            return ring[:n]
        at = self._n % self._size
        return np.concatenate([ring[at:], ring[:at]])

    def _column(self, name):
        return name if self.prefix in name else self.prefix + name

    # ---- updates
    def push(self, values, carry=None):
        """Record one mini-batch.  ``carry``: means to stand in for keys this
        mini-batch did not compute (a network that sat the batch out)."""
        row = {self._column(k): float(v) for k, v in values.items()}
        if carry:
            for k, v in carry.items():
                if k.startswith(self.prefix) and k not in row:
                    row[k] = float(v)
        at = self._n % self._size
        for k in self._rows:
            self._rows[k][at] = np.nan
        for k, v in row.items():
            if k not in self._rows:
                self._rows[k] = np.full(self._size, np.nan)
            self._rows[k][at] = v
        self._n += 1

    def seed(self, means):
        """Start from the last epoch of a loaded history (one row)."""
        row = {k: float(v) for k, v in means.items()
               if self.prefix in k and np.isfinite(float(v))}
        if row:
            self.push(row)

    # ---- views
    def means(self):
        out = {}
        # pandas ``mean`` semantics (skipna): only NaN is dropped — a
        # diverged batch (inf) stays visible in the running mean that drives
        # the train / skip gating (base.py:1161-1164)
        for k, ring in self._rows.items():
            vals = ring[~np.isnan(ring)]
            if len(vals):
                with np.errstate(invalid='ignore'):
                    out[k] = float(vals.mean())
        return out

    def last(self, key, default=None):
        key = self._column(key)
        if key not in self._rows or self._n == 0:
            return default
        v = self._rows[key][(self._n - 1) % self._size]
        return default if not np.isfinite(v) else float(v)

    def __len__(self):
        return min(self._n, self._size)


class History:
    """Epoch table; ``.frame`` is the ``model.history`` DataFrame (index
    ``epoch``), written to ``history.csv`` by ``save``."""

    def __init__(self, frame=None):
        if isinstance(frame, str):
            frame = pd.read_csv(frame, index_col=0)
        if frame is None:
            self._rows, self._index = [], []
            self._frame = None
        else:
            self._frame = frame
            self._rows, self._index = None, None

    @property
    def started(self):
        return self._frame is not None or bool(self._rows)

    def next_epochs(self, n):
        first = int(self.frame.index.values[-1]) + 1 if self.started and \
            len(self.frame) else 0
        return list(range(first, first + n))

    def last_row(self):
        f = self.frame
        return {} if f is None or not len(f) else f.iloc[-1].to_dict()

    def write(self, epoch, values):
        """Add / extend the row of ``epoch``."""
        f = self.frame
        if f is None:
            f = pd.DataFrame(columns=['elapsed_time'])
            f.index.name = 'epoch'
        new = [k for k in values if k not in f.columns]
        if new:
            f = pd.concat([f, pd.DataFrame(columns=new, index=f.index,
                                           dtype=float)], axis=1)
        if epoch not in f.index:
            f.loc[epoch] = np.nan
        f.loc[epoch, list(values)] = [values[k] for k in values]
        f.index.name = 'epoch'
        self._frame = f

    @property
    def frame(self):
        return self._frame

    def plateaued(self, column, threshold, n_epoch):
        """True once the last ``n_epoch`` epoch-to-epoch changes of ``column``
        are all below ``threshold`` (more than ``n_epoch + 1`` epochs seen)."""
        f = self.frame
        if f is None or column not in f or len(f) <= n_epoch + 1:
            return False, None
        steps = np.abs(np.diff(np.asarray(f[column], dtype=np.float64)))
        tail = steps[-n_epoch:]
        return bool((tail < threshold).all()), tail
