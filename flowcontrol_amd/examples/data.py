"""Input data of the example cases: the reference's shipped mesh and controller DATA files (``src/examples/*/data_input``),
converted once to ``.npz`` by ``tests/golden/make_mesh_fixtures.py`` (meshes; the ``.mat`` controller is the reference's file as it
is).  They live next to the cases that use them, as in the reference; tests and benchmarks read them from here too."""
from __future__ import annotations

from pathlib import Path

HERE = Path(__file__).resolve().parent

_MESHES = {
    "O1": "cylinder", "cavity_coarse": "cavity", "cavity_fine": "cavity", "mesh_middle_gmsh": "pinball", "lidcavity_mesh64": "lidcavity",
}
_CONTROLLERS = {"Kopt_reduced13": "cylinder"}


def mesh_file(name: str) -> Path:
    """``<case>/data_input/<name>.npz`` of a shipped mesh (``O1``, ``cavity_coarse``, ``cavity_fine``, ``mesh_middle_gmsh``, ``lidcavity_mesh64``)."""
    name = name[:-4] if name.endswith(".npz") else name
    if name not in _MESHES:
        raise KeyError(f"no shipped mesh {name!r} (have {sorted(_MESHES)})")
    return HERE / _MESHES[name] / "data_input" / f"{name}.npz"


def controller_file(name: str = "Kopt_reduced13") -> Path:
    """``<case>/data_input/<name>.mat`` of a shipped controller (MATLAB v5: A, B, C, D)."""
    name = name[:-4] if name.endswith(".mat") else name
    if name not in _CONTROLLERS:
        raise KeyError(f"no shipped controller {name!r} (have {sorted(_CONTROLLERS)})")
    return HERE / _CONTROLLERS[name] / "data_input" / f"{name}.mat"


__all__ = ["mesh_file", "controller_file"]
