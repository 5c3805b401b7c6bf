"""Pick / Detection records returned by ``classify`` (seisbench.util.annotations
equivalents: fields and printed form as in README.md:69-78, Final_models/demo.ipynb:410-413)."""
from __future__ import annotations

from collections import UserList
from dataclasses import dataclass, field
from typing import Any, Optional


@dataclass
class Pick:
    trace_id: str
    start_time: Any
    end_time: Any = None
    peak_time: Any = None
    peak_value: Optional[float] = None
    phase: Optional[str] = None

    def _sort_key(self):
        return (self.start_time, self.trace_id, self.phase or "")

    def __lt__(self, other):
        return self._sort_key() < other._sort_key()

    def __str__(self):
        parts = [self.trace_id]
        parts.append(str(self.peak_time if self.peak_time is not None else self.start_time))
        if self.phase is not None:
            parts.append(str(self.phase))
        return "\t".join(parts)


@dataclass
class Detection:
    trace_id: str
    start_time: Any
    end_time: Any
    peak_value: Optional[float] = None

    def __lt__(self, other):
        return (self.start_time, self.trace_id) < (other.start_time, other.trace_id)

    def __str__(self):
        return "\t".join([self.trace_id, str(self.start_time), str(self.end_time)])


class _PrintableList(UserList):
    """A list of records (``collections.UserList``, as seisbench.util.annotations.PickList).  ``classify`` hands its
    triggers over as sorted COLUMNS plus a function that builds the records; the records come into being at the first
    access of ``.data`` (any list operation but ``len``).  A station-day holds ~2,000 records of five objects each: built
    eagerly inside every call they drove CPython's cyclic collector into a full collection every ~12 calls, a 30-37 ms pause
    in a 24 ms call (``tools/api_outlier.py``); a caller that only counts, or that takes the records once, does not pay for
    them inside the call."""

    _name = "List"

    def __init__(self, initlist=None):
        self._lazy = None
        self._data = []
        super().__init__(initlist)

    @classmethod
    def _deferred(cls, n, make):
        """``n`` records that ``make()`` (-> list) builds on first access."""
        out = cls()
        if n:
            out._lazy = (int(n), make)
        return out

    @property
    def data(self):
        if self._lazy is not None:
            (_, make), self._lazy = self._lazy, None
            self._data = list(make())
        return self._data

    @data.setter
    def data(self, value):
        self._lazy = None
        self._data = value

    def __len__(self):
        return self._lazy[0] if self._lazy is not None else len(self._data)

    def __copy__(self):  # UserList.__copy__ reads self.__dict__["data"]; here `data` is a property
        return self.__class__(self.data)

    def __getstate__(self):  # pickling / copying hands over the records, never the deferred builder
        return {"_data": self.data, "_lazy": None}

    def __setstate__(self, state):
        self._lazy = None
        self._data = list(state["_data"])

    def copy(self):
        return self.__class__(self.data)

    def __str__(self):
        head = f"{self._name} with {len(self)} entries:\n\n"
        if len(self) <= 20:
            return head + "\n".join(str(x) for x in self)
        return head + "\n".join(str(x) for x in self[:5]) + "\n...\n" + "\n".join(str(x) for x in self[-5:])

    __repr__ = __str__

    def select(self, trace_id=None, min_confidence=None, phase=None):
        import re

        out = self.__class__()
        for x in self:
            if trace_id is not None and not re.fullmatch(trace_id, x.trace_id):
                continue
            if min_confidence is not None and (x.peak_value is None or x.peak_value < min_confidence):
                continue
            if phase is not None and getattr(x, "phase", None) != phase:
                continue
            out.append(x)
        return out


class PickList(_PrintableList):
    _name = "PickList"


class DetectionList(_PrintableList):
    _name = "DetectionList"


@dataclass
class ClassifyOutput:
    creator: str
    picks: PickList = field(default_factory=PickList)
    detections: DetectionList = field(default_factory=DetectionList)

    def __str__(self):
        return f"ClassifyOutput(creator={self.creator}, picks={len(self.picks)}, detections={len(self.detections)})"
