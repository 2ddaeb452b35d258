"""Multi-indices and coefficient tables of linear differential operators.

Host mirror of `linfuncops/diffops/_coefficients.py:9-198` (same public names and
semantics: `{codomain_index: {MultiIndex: coefficient}}`, `+`, unary `-`, scalar `*`,
shape validation), written independently.
"""

from __future__ import annotations

from collections.abc import Mapping

import numpy as np


class MultiIndex:
    """Orders of a partial derivative, one non-negative integer per input entry."""

    __slots__ = ("_a",)

    def __init__(self, multi_index):
        a = np.array(multi_index, dtype=int)
        if (a < 0).any():
            raise ValueError(f"Multi-index {multi_index} contains negative entries.")
        a.setflags(write=False)
        self._a = a

    @classmethod
    def from_index(cls, index, shape, order: int) -> "MultiIndex":
        a = np.zeros(shape, dtype=int)
        a[index] = order
        return cls(a)

    @property
    def array(self) -> np.ndarray:
        return self._a

    @property
    def shape(self):
        return self._a.shape

    @property
    def order(self) -> int:
        return int(self._a.sum())

    @property
    def is_mixed(self) -> bool:
        return int(np.count_nonzero(self._a)) > 1

    def as_tuple(self) -> tuple[int, ...]:
        return tuple(int(v) for v in self._a.reshape(-1))

    def __getitem__(self, index):
        return self._a[index]

    def __hash__(self):
        return hash((self._a.shape, self.as_tuple()))

    def __eq__(self, other):
        if not isinstance(other, MultiIndex):
            return NotImplemented
        return self._a.shape == other._a.shape and bool((self._a == other._a).all())

    def __repr__(self):
        return f"MultiIndex({self._a.tolist()})"


class PartialDerivativeCoefficients(Mapping):
    """`{codomain_index: {MultiIndex: coefficient}}` of an operator  sum_a c_a d^a."""

    def __init__(self, coefficient_dict, input_domain_shape, input_codomain_shape):
        input_domain_shape = tuple(input_domain_shape)
        input_codomain_shape = tuple(input_codomain_shape)
        count = 0
        for codomain_idx, entries in coefficient_dict.items():
            ok = len(codomain_idx) == len(input_codomain_shape) and all(
                0 <= i < s for i, s in zip(codomain_idx, input_codomain_shape))
            if not ok:
                raise ValueError(
                    f"Codomain index {codomain_idx} does not match shape {input_codomain_shape}.")
            for mi in entries:
                if tuple(mi.shape) != input_domain_shape:
                    raise ValueError(
                        f"Multi-index shape {mi.shape} does not match input domain shape "
                        f"{input_domain_shape}.")
                count += 1
        self._d = {ci: dict(e) for ci, e in coefficient_dict.items()}
        self._num_entries = count
        self._input_domain_shape = input_domain_shape
        self._input_codomain_shape = input_codomain_shape

    @property
    def num_entries(self) -> int:
        return self._num_entries

    @property
    def has_mixed(self) -> bool:
        return any(mi.is_mixed for e in self._d.values() for mi in e)

    @property
    def input_domain_shape(self):
        return self._input_domain_shape

    @property
    def input_codomain_shape(self):
        return self._input_codomain_shape

    def __getitem__(self, codomain_idx):
        return self._d[codomain_idx]

    def __len__(self):
        return len(self._d)

    def __iter__(self):
        return iter(self._d)

    def __neg__(self):
        return -1.0 * self

    def __add__(self, other):
        if not isinstance(other, PartialDerivativeCoefficients):
            return NotImplemented
        if self.input_domain_shape != other.input_domain_shape:
            raise ValueError(
                "Cannot add coefficients with input domain shapes "
                f"{self.input_domain_shape} != {other.input_domain_shape}")
        if self.input_codomain_shape != other.input_codomain_shape:
            raise ValueError(
                "Cannot add coefficients with input codomain shapes "
                f"{self.input_codomain_shape} != {other.input_codomain_shape}")
        merged = {ci: dict(e) for ci, e in self._d.items()}
        for ci, entries in other.items():
            tgt = merged.setdefault(ci, {})
            for mi, c in entries.items():
                tgt[mi] = tgt.get(mi, 0.0) + c
        return PartialDerivativeCoefficients(merged, self.input_domain_shape, self.input_codomain_shape)

    def __sub__(self, other):
        return self + (-other)

    def __rmul__(self, other):
        if np.ndim(other) != 0:
            return NotImplemented
        scaled = {ci: {mi: other * c for mi, c in e.items()} for ci, e in self._d.items()}
        return PartialDerivativeCoefficients(scaled, self.input_domain_shape, self.input_codomain_shape)
