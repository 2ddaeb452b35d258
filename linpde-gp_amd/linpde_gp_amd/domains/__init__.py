"""Domains and grids -- host mirror of `linpde_gp.domains`
(`domains/_interval.py:14-83`, `_box.py:16-114`, `_point.py:12`, `_cartesian_product.py:16-127`)
and of `TensorProductGrid` (`randprocs/covfuncs/_tensor_product.py:133-149`).
Grid generators only: `uniform_grid(shape, inset)` is `linspace` per factor +
`meshgrid(indexing="ij")`; `boundary` enumerates the faces factor by factor."""

from __future__ import annotations

import numpy as np


class TensorProductGrid(np.ndarray):
    """`meshgrid(*factors, indexing="ij")` stacked on the last axis; remembers its factors."""

    def __new__(cls, *factors, indexing="ij"):
        factors = tuple(np.asarray(f, dtype=np.double) for f in factors)
        obj = np.stack(np.meshgrid(*factors, copy=True, sparse=False, indexing=indexing), axis=-1).view(cls)
        obj.factors = factors
        return obj

    def __array_finalize__(self, obj):
        self.factors = getattr(obj, "factors", None)

    # pickling (the single-process multi-GPU front ships observation points to its workers): keep the factors
    def __reduce__(self):
        fn, args, state = super().__reduce__()
        return fn, args, (state, self.factors)

    def __setstate__(self, state):
        base, factors = state
        super().__setstate__(base)
        self.factors = factors


class Domain:
    def __init__(self, shape, dtype=np.double):
        self._shape = tuple(shape)
        self._dtype = np.dtype(dtype)

    @property
    def shape(self):
        return self._shape

    @property
    def ndims(self):
        return len(self._shape)

    @property
    def size(self):
        return int(np.prod(self._shape, dtype=int))

    @property
    def dtype(self):
        return self._dtype


class Point(Domain):
    def __init__(self, point):
        self._point = np.asarray(point, dtype=np.double)
        super().__init__(self._point.shape)

    @property
    def boundary(self):
        return (self,)

    @property
    def volume(self):
        return 0.0

    def __array__(self, dtype=None, copy=None):
        return np.array(self._point, dtype=dtype)

    def __float__(self):
        return float(self._point)

    def __contains__(self, item):
        a = np.asarray(item, dtype=np.double)
        return a.shape == self.shape and bool(np.all(a == self._point))

    def __eq__(self, other):
        return isinstance(other, Point) and self.shape == other.shape and bool(np.all(self._point == other._point))

    def __hash__(self):
        return hash(self._point.tobytes())

    def __repr__(self):
        return f"<Point {self._point} with shape={self.shape}>"


class Interval(Domain):
    def __init__(self, lower_bound, upper_bound):
        self._lower_bound = float(lower_bound)
        self._upper_bound = float(upper_bound)
        if self._lower_bound > self._upper_bound:
            raise ValueError("The lower bound must not be larger than the upper bound")
        super().__init__(())

    def __len__(self):
        return 2

    def __getitem__(self, idx):
        if idx in (0, -2):
            return self._lower_bound
        if idx in (1, -1):
            return self._upper_bound
        raise KeyError(f"Index {idx} is out of range")

    def __iter__(self):
        yield self._lower_bound
        yield self._upper_bound

    @property
    def boundary(self):
        return (Point(self._lower_bound), Point(self._upper_bound))

    @property
    def volume(self):
        return self._upper_bound - self._lower_bound

    def __contains__(self, item):
        a = np.asarray(item, dtype=np.double)
        return a.shape == () and self._lower_bound <= a <= self._upper_bound

    def __eq__(self, other):
        return isinstance(other, Interval) and tuple(self) == tuple(other)

    def __hash__(self):
        return hash(tuple(self))

    def uniform_grid(self, shape, inset=0.0) -> np.ndarray:
        shape = (int(shape),) if np.ndim(shape) == 0 else tuple(int(s) for s in shape)
        inset = np.asarray(inset, dtype=np.double)
        if len(shape) != 1 or inset.ndim != 0:
            raise ValueError("`Interval.uniform_grid` needs a one-dimensional shape and a scalar inset")
        return np.linspace(self._lower_bound + inset, self._upper_bound - inset, shape[0])

    def __repr__(self):
        return f"<Interval {[self._lower_bound, self._upper_bound]}>"


def asdomain(arg) -> Domain:
    if isinstance(arg, Domain):
        return arg
    a = np.asarray(arg, dtype=np.double)
    if a.ndim == 0:
        return Point(a)
    if a.shape == (2,):
        return Interval(a[0], a[1])
    if a.ndim == 2 and a.shape[-1] == 2:
        return Box(a)
    raise TypeError(f"Could not convert {arg!r} to a domain")


class CartesianProduct(Domain):
    def __init__(self, *domains):
        self._domains = tuple(asdomain(d) for d in domains)
        if not all(d.ndims <= 1 for d in self._domains):
            raise ValueError("factors must be scalar or vector domains")
        super().__init__((sum(max(d.size, 1) for d in self._domains),))

    @property
    def factors(self):
        return self._domains

    def __len__(self):
        return len(self._domains)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return self._domains[idx]
        return CartesianProduct(*self._domains[idx])

    def __iter__(self):
        return iter(self._domains)

    @property
    def boundary(self):
        """Faces: every factor replaced in turn by each part of its boundary
        (`_cartesian_product.py:75-82`)."""
        return tuple(
            CartesianProduct(*self._domains[:i], part, *self._domains[i + 1:])
            for i, f in enumerate(self._domains)
            for part in f.boundary
        )

    @property
    def volume(self):
        v = 1.0
        for d in self._domains:
            v *= d.volume
        return v

    def _flat_bounds(self):
        bounds = []
        for d in self._domains:
            if isinstance(d, Interval):
                bounds.append(tuple(d))
            elif isinstance(d, Point):
                for c in np.atleast_1d(np.asarray(d)):
                    bounds.append((float(c), float(c)))
            elif isinstance(d, Box):
                bounds.extend((float(lo), float(hi)) for lo, hi in d.bounds)
            else:
                raise NotImplementedError(f"no box form for factor {d!r}")
        return np.array(bounds, dtype=np.double)

    def uniform_grid(self, shape, inset=0.0):
        return Box(self._flat_bounds()).uniform_grid(shape, inset=inset)

    def __eq__(self, other):
        return isinstance(other, CartesianProduct) and len(self) == len(other) and all(
            a == b for a, b in zip(self._domains, other._domains))

    def __hash__(self):
        return hash(self._domains)

    def __repr__(self):
        return "<CartesianProduct of " + ", ".join(repr(d) for d in self._domains) + ">"


class Box(CartesianProduct):
    def __init__(self, bounds):
        b = np.array(bounds, dtype=np.double, copy=True)
        if not (b.ndim == 2 and b.shape[-1] == 2):
            raise ValueError(f"`bounds` must have shape (D, 2), but an object of shape {b.shape} was given.")
        if not np.all(b[:, 0] <= b[:, 1]):
            raise ValueError("The lower bounds must not be larger than the upper bounds.")
        b.flags.writeable = False
        self._bounds = b
        self._interior_idcs = np.nonzero(b[:, 0] != b[:, 1])[0]
        super().__init__(*(Interval(lo, hi) if lo != hi else Point(lo) for lo, hi in b))

    @property
    def bounds(self):
        return self._bounds

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return self._domains[idx]
        return Box(self._bounds[idx, :])

    def __contains__(self, item):
        a = np.asarray(item, dtype=np.double)
        return a.shape == self.shape and bool(np.all((self._bounds[:, 0] <= a) & (a <= self._bounds[:, 1])))

    def uniform_grid(self, shape, inset=0.0) -> TensorProductGrid:
        """`shape`/`inset` refer to the non-collapsed dimensions (`_box.py:81-114`)."""
        k = len(self._interior_idcs)
        shape = (int(shape),) * k if np.ndim(shape) == 0 else tuple(int(s) for s in shape)
        if len(shape) != k:
            raise ValueError(f"expected a shape with {k} entries (non-collapsed dimensions)")
        insets = np.broadcast_to(np.asarray(inset, dtype=np.double), (k,))
        nums = np.ones(len(self._bounds), dtype=int)
        ins = np.zeros(len(self._bounds))
        nums[self._interior_idcs] = shape
        ins[self._interior_idcs] = insets
        factors = []
        for (lo, hi), n, i in zip(self._bounds, nums, ins):
            factors.append(np.linspace(lo + i, hi - i, n) if lo != hi else np.array([lo]))
        return TensorProductGrid(*factors, indexing="ij")

    def __repr__(self):
        return f"<Box {self._bounds.tolist()}>"


__all__ = ["Domain", "Point", "Interval", "Box", "CartesianProduct", "TensorProductGrid", "asdomain"]
