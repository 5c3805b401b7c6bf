"""Minimal waveform containers with the ObsPy surface the picker API touches.

ObsPy is not a dependency of this package (it is absent from the build image).
``Stream``/``Trace``/``UTCDateTime`` below reproduce the attributes the
reference's usage reads (README.md:38-82, Final_models/demo.ipynb:242-327):
``trace.stats.{network,station,location,channel,starttime,sampling_rate,npts,
endtime,delta}``, ``trace.data``, ``trace.id``, ``trace.times(reftime=...)``,
``stream.select(channel=...)``, ``stream.copy()``, ``stream.merge()``,
iteration / indexing / ``len``.  Real ObsPy streams are accepted too (duck
typing); see ``models._group_stream``.
"""
from __future__ import annotations

import copy as _copy
import fnmatch
import math
from datetime import datetime, timedelta, timezone

import numpy as np

_EPOCH = datetime(1970, 1, 1, tzinfo=timezone.utc)


def pinned_array(shape, dtype=np.float32):
    """A numpy array in page-locked host memory (it keeps the torch tensor that owns the pages alive).  Trace data that lives
    in such arrays is uploaded by DMA straight from the caller's pages -- no staging copy through the runtime's own pinned
    buffers -- and asynchronously, so ``classify()`` overlaps the upload of the next station with the current one's compute at
    the link's full rate.  (Page-locking costs ~0.1 ms per MB once: for buffers that are filled many times -- a reader's
    ring, a real-time feed -- not for arrays used once.)"""
    import torch

    return torch.empty(tuple(np.atleast_1d(shape)), dtype=getattr(torch, np.dtype(dtype).name)).pin_memory().numpy()


class UTCDateTime:
    """UTC time stamp with microsecond resolution (integer microseconds since the epoch)."""

    __slots__ = ("_us",)

    def __init__(self, value=0.0):
        if isinstance(value, UTCDateTime):
            self._us = value._us
        elif isinstance(value, (int, float, np.integer, np.floating)):
            self._us = int(round(float(value) * 1e6))
        elif isinstance(value, datetime):
            if value.tzinfo is None:
                value = value.replace(tzinfo=timezone.utc)
            d = value - _EPOCH
            self._us = (d.days * 86400 + d.seconds) * 1_000_000 + d.microseconds
        elif isinstance(value, str):
            s = value.strip().replace("Z", "")
            fmt = "%Y-%m-%dT%H:%M:%S.%f" if "." in s else "%Y-%m-%dT%H:%M:%S"
            if "T" not in s:
                fmt = fmt.replace("T", " ") if " " in s else "%Y-%m-%d"
            self.__init__(datetime.strptime(s, fmt))
        elif hasattr(value, "timestamp"):  # obspy.UTCDateTime
            ts = value.timestamp
            self._us = int(round(float(ts() if callable(ts) else ts) * 1e6))
        else:
            raise TypeError(f"cannot build UTCDateTime from {type(value)}")

    @classmethod
    def _from_us(cls, us):
        o = cls.__new__(cls)
        o._us = int(us)
        return o

    @property
    def timestamp(self) -> float:
        return self._us / 1e6

    @property
    def datetime(self) -> datetime:
        return _EPOCH + timedelta(microseconds=self._us)

    def __add__(self, seconds):
        return UTCDateTime._from_us(self._us + int(round(float(seconds) * 1e6)))

    __radd__ = __add__

    def __sub__(self, other):
        if isinstance(other, UTCDateTime):
            return (self._us - other._us) / 1e6
        if hasattr(other, "timestamp") and not isinstance(other, (int, float)):
            return self.timestamp - UTCDateTime(other).timestamp
        return UTCDateTime._from_us(self._us - int(round(float(other) * 1e6)))

    def _key(self, other):
        return UTCDateTime(other)._us

    def __eq__(self, other):
        try:
            return self._us == self._key(other)
        except TypeError:
            return NotImplemented

    def __lt__(self, other):
        return self._us < self._key(other)

    def __le__(self, other):
        return self._us <= self._key(other)

    def __gt__(self, other):
        return self._us > self._key(other)

    def __ge__(self, other):
        return self._us >= self._key(other)

    def __hash__(self):
        return hash(self._us)

    def __float__(self):
        return self.timestamp

    def __str__(self):
        return self.datetime.strftime("%Y-%m-%dT%H:%M:%S.%f") + "Z"

    def __repr__(self):
        return f"UTCDateTime({str(self)!r})"


class Stats(dict):
    """Trace header; attribute and item access, derived npts / delta / endtime."""

    _defaults = dict(network="", station="", location="", channel="", starttime=None, sampling_rate=1.0, npts=0)

    def __init__(self, header=None):
        super().__init__(self._defaults)
        self["starttime"] = UTCDateTime(0)
        if header:
            for k, v in dict(header).items():
                self[k] = v

    def __setitem__(self, k, v):
        if k == "starttime":
            v = UTCDateTime(v)
        elif k == "sampling_rate":
            v = float(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        if k == "delta":
            return 1.0 / self["sampling_rate"]
        if k == "endtime":
            return self["starttime"] + max(self["npts"] - 1, 0) / self["sampling_rate"]
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return Stats(dict(self))


class Trace:
    """``data`` is a numpy array.  A trace may instead be backed by a device array (``device_data``: a CUDA
    torch tensor, as ``volpick_amd.read(..., device_resident=True)`` produces); ``data`` then materialises a
    host copy on first access, and the picker consumes the device array directly."""

    def __init__(self, data=None, header=None, device_data=None):
        self._dev = device_data
        if data is None and device_data is not None:
            self._data = None
            n = int(device_data.shape[0])
        else:
            self._data = np.asarray(data if data is not None else np.zeros(0, dtype=np.float32))
            n = len(self._data)
        self.stats = Stats(header)
        self.stats["npts"] = n

    @property
    def data(self):
        if self._data is None:
            self._data = self._dev.cpu().numpy()
        return self._data

    @data.setter
    def data(self, value):
        self._data = np.asarray(value)
        self._dev = None
        self.stats["npts"] = len(self._data)

    @property
    def id(self):
        s = self.stats
        return f"{s.network}.{s.station}.{s.location}.{s.channel}"

    def times(self, type="relative", reftime=None):
        t = np.arange(len(self.data)) / self.stats.sampling_rate
        if reftime is not None:
            t = t + (self.stats.starttime - UTCDateTime(reftime))
        return t

    def copy(self):
        return Trace(self.data.copy(), self.stats.copy())

    def __len__(self):
        return int(self.stats["npts"])

    def __str__(self):
        s = self.stats
        return f"{self.id} | {s.starttime} - {s.endtime} | {s.sampling_rate:.1f} Hz, {len(self)} samples"

    __repr__ = __str__


class Stream:
    def __init__(self, traces=None):
        self.traces = list(traces) if traces is not None else []

    def __iter__(self):
        return iter(self.traces)

    def __len__(self):
        return len(self.traces)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return Stream(self.traces[i])
        return self.traces[i]

    def __add__(self, other):
        return Stream(self.traces + list(other))

    def __iadd__(self, other):
        self.traces.extend(list(other))
        return self

    def append(self, tr):
        self.traces.append(tr)
        return self

    def copy(self):
        return Stream([t.copy() for t in self.traces])

    def sort(self, keys=("network", "station", "location", "channel", "starttime")):
        self.traces.sort(key=lambda t: tuple(str(t.stats[k]) if k != "starttime" else t.stats[k]._us for k in keys))
        return self

    def select(self, network=None, station=None, location=None, channel=None, component=None, id=None):
        out = []
        for tr in self.traces:
            s = tr.stats
            if network is not None and not fnmatch.fnmatchcase(s.network, network):
                continue
            if station is not None and not fnmatch.fnmatchcase(s.station, station):
                continue
            if location is not None and not fnmatch.fnmatchcase(s.location, location):
                continue
            if channel is not None and not fnmatch.fnmatchcase(s.channel, channel):
                continue
            if component is not None and not (s.channel and fnmatch.fnmatchcase(s.channel[-1], component)):
                continue
            if id is not None and not fnmatch.fnmatchcase(tr.id, id):
                continue
            out.append(tr)
        return Stream(out)

    def merge(self, method=0, fill_value=None):
        """Join traces of the same id.  ``method=-1`` (the clean-up merge annotate() performs)
        only joins exactly contiguous / duplicate-free pieces; other methods additionally fill
        gaps with ``fill_value`` (0 if None).  Overlaps keep the earlier trace's samples."""
        groups = {}
        for tr in self.traces:
            groups.setdefault((tr.id, tr.stats.sampling_rate), []).append(tr)
        merged = []
        for (_, sr), trs in groups.items():
            trs.sort(key=lambda t: t.stats.starttime._us)
            cur = trs[0].copy()
            for nxt in trs[1:]:
                expected = cur.stats.starttime + len(cur.data) / sr
                gap = int(round((nxt.stats.starttime - expected) * sr))
                if gap == 0:
                    cur.data = np.concatenate([cur.data, nxt.data])
                elif gap > 0 and method != -1:
                    fill = np.full(gap, 0 if fill_value is None else fill_value, dtype=cur.data.dtype)
                    cur.data = np.concatenate([cur.data, fill, nxt.data])
                elif gap < 0 and method != -1:
                    keep = nxt.data[-gap:] if -gap < len(nxt.data) else nxt.data[:0]
                    cur.data = np.concatenate([cur.data, keep])
                else:
                    cur.stats["npts"] = len(cur.data)
                    merged.append(cur)
                    cur = nxt.copy()
                    continue
            cur.stats["npts"] = len(cur.data)
            merged.append(cur)
        self.traces = merged
        return self

    def __str__(self):
        return f"{len(self.traces)} Trace(s) in Stream:\n" + "\n".join(str(t) for t in self.traces)

    __repr__ = __str__


def deepcopy_stream(stream):
    return stream.copy() if hasattr(stream, "copy") else _copy.deepcopy(stream)


__all__ = ["UTCDateTime", "Stats", "Trace", "Stream", "math"]
