"""Pick / Detection records returned by ``classify`` (seisbench.util.annotations
equivalents: fields and printed form as in README.md:69-78, Final_models/demo.ipynb:410-413)."""
from __future__ import annotations

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


class _PrintableList(list):
    _name = "List"

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
