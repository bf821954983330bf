"""Force models of the reference (src/repel_forces.jl): names, defaults and constructors mirror
the Julia structs; `compute_force` is the host-side scalar (the device evaluates the same laws
inside the sweep, csrc/wtp_device.hpp force_law)."""
from __future__ import annotations

from dataclasses import dataclass


class RepelForceModel:
    kind = -1

    def desc(self) -> dict:
        raise NotImplementedError


@dataclass(frozen=True)
class InverseDistanceForce(RepelForceModel):  # src/repel_forces.jl:31-37
    beta: float = 0.2
    kind = 0

    def desc(self):
        return dict(kind=0, beta=self.beta, u0=1.0, gamma=3.0)


@dataclass(frozen=True)
class SpacingEquilibriumForce(RepelForceModel):  # :51-60
    beta: float = 0.2
    kind = 1

    def desc(self):
        return dict(kind=1, beta=self.beta, u0=1.0, gamma=3.0)


@dataclass(frozen=True)
class ClippedSpacingForce(RepelForceModel):  # :88-100 (the default law)
    beta: float = 0.2
    u0: float = 1.0
    kind = 2

    def desc(self):
        return dict(kind=2, beta=self.beta, u0=self.u0, gamma=3.0)


@dataclass(frozen=True)
class StrongSpacingForce(RepelForceModel):  # :116-127
    beta: float = 0.2
    gamma: float = 3.0
    kind = 3

    def desc(self):
        return dict(kind=3, beta=self.beta, u0=1.0, gamma=self.gamma)


def compute_force(model: RepelForceModel, u: float) -> float:
    """F(u), u = r/s (src/repel_forces.jl:37,57-60,96-100,124-127)."""
    u2 = u * u
    if isinstance(model, InverseDistanceForce):
        return 1.0 / (u2 + model.beta) ** 2
    if isinstance(model, SpacingEquilibriumForce):
        return (1 - u2) / (u2 + model.beta) ** 2
    if isinstance(model, ClippedSpacingForce):
        f = (model.u0 * model.u0 - u2) / (u2 + model.beta) ** 2
        return max(f, 0.0)
    if isinstance(model, StrongSpacingForce):
        return (1 - u2) / (u2 + model.beta) ** model.gamma
    raise TypeError(f"not a RepelForceModel: {model!r}")
