"""State-space operators of the linearised flow, ``E ẋ = A x + B u, y = C x`` — the reference's
``src/examples/operators/compute_operators.py``: base flow, then ``OperatorGetter.get_all()`` and the files ``A.npz`` / ``A_coo.npz`` /
``E.npz`` / ``E_coo.npz`` (scipy sparse) that the control-design scripts read.  The Jacobian is assembled by the HIP element loop
(``fc_assemble_matrix``), downloaded once.

    python -m flowcontrol_amd.examples.operators.compute_operators [cylinder|cavity|lidcavity|pinball] [out_dir]
"""
import logging
import sys
from pathlib import Path

from flowcontrol_amd.io import export_square_operators
from flowcontrol_amd.operatorgetter import OperatorGetter

logger = logging.getLogger(__name__)


def compute_operators_flowsolver(flowsolver, export, path=None):
    logger.info("Now computing operators...")
    A, E, B, C = OperatorGetter(flowsolver).get_all()
    if export:
        export_square_operators(path=Path(path) if path else Path.cwd() / "data_output", operators=[A, E], operators_names=["A", "E"])
    return A, E, B, C


def base_flow(case: str, path_out: Path):
    """The base-flow recipe the reference script uses for each case."""
    if case == "cylinder":
        from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver

        fs = CylinderFlowSolver.make_default(Re=100, path_out=path_out)
        fs.compute_steady_state(method="picard", max_iter=3, tol=1e-7, u_ctrl=[0.0, 0.0])
        fs.compute_steady_state(method="newton", max_iter=25, u_ctrl=[0.0, 0.0], initial_guess=fs.fields.UP0)
    elif case == "cavity":
        from flowcontrol_amd.examples.cavity.cavityflowsolver import CavityFlowSolver

        fs = CavityFlowSolver.make_default(Re=7500, path_out=path_out)
        fs.compute_steady_state(method="picard", max_iter=10, tol=1e-7, u_ctrl=[0.0])
        fs.compute_steady_state(method="newton", max_iter=10, u_ctrl=[0.0], initial_guess=fs.fields.UP0)
    elif case == "lidcavity":
        from flowcontrol_amd.examples.lidcavity.lidcavityflowsolver import LidCavityFlowSolver

        fs = LidCavityFlowSolver.make_default(Re=1000, path_out=path_out)
        fs.compute_steady_state(method="picard", max_iter=40, tol=1e-7, u_ctrl=[0.0])
    elif case == "pinball":
        from flowcontrol_amd.examples.pinball.pinballflowsolver import PinballFlowSolver

        fs = PinballFlowSolver.make_default(Re=50, path_out=path_out)
        fs.compute_steady_state(method="newton", max_iter=25, u_ctrl=[0.0, 0.0, 0.0])
    else:
        raise ValueError(f"unknown case {case!r}")
    return fs


def main(case: str = "cylinder", out: Path | None = None):
    logging.basicConfig(level=logging.INFO)
    out = Path(out) if out else Path.cwd() / "data_output" / case
    fs = base_flow(case, out)
    A, E, B, C = compute_operators_flowsolver(fs, export=True, path=out)
    print(f"{case}: A {A.shape} nnz {A.nnz}, E nnz {E.nnz}, B {B.shape}, C {C.shape} -> {out}")
    return A, E, B, C


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "cylinder", Path(sys.argv[2]) if len(sys.argv) > 2 else None)
