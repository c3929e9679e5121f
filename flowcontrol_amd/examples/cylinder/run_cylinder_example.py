"""Flow past a cylinder, closed loop + restart — the reference's
``src/examples/cylinder/run_cylinder_example.py`` on an MI355X (same sequence of calls).

    python -m flowcontrol_amd.examples.cylinder.run_cylinder_example [num_steps]
"""

import logging
import sys
from pathlib import Path

import numpy as np

from flowcontrol_amd import flowsolverparameters
from flowcontrol_amd import utils as flu
from flowcontrol_amd.actuator import ActuatorBCParabolicV
from flowcontrol_amd.controller import Controller
from flowcontrol_amd.examples.cylinder.cylinderflowsolver import DEFAULT_MESH, CylinderFlowSolver
from flowcontrol_amd.sensor import SENSOR_TYPE, SensorPoint

logging.basicConfig(level=logging.INFO)
GOLDEN = Path(__file__).resolve().parents[3] / "tests" / "golden"


def make_params(path_out, num_steps, save_every, Tstart=0.0):
    params_flow = flowsolverparameters.ParamFlow(Re=100, uinf=1.0)
    params_flow.user_data["D"] = 1.0
    params_time = flowsolverparameters.ParamTime(num_steps=num_steps, dt=0.005, Tstart=Tstart)
    params_save = flowsolverparameters.ParamSave(save_every=save_every, path_out=path_out)
    params_solver = flowsolverparameters.ParamSolver(throw_error=True, is_eq_nonlinear=True, shift=0.0)
    params_mesh = flowsolverparameters.ParamMesh(meshpath=DEFAULT_MESH)
    params_mesh.user_data.update({"xinf": 20, "xinfa": -10, "yinf": 10})
    width = ActuatorBCParabolicV.angular_size_deg_to_width(10, params_flow.user_data["D"] / 2)
    params_control = flowsolverparameters.ParamControl(
        sensor_list=[SensorPoint(sensor_type=SENSOR_TYPE.V, position=np.array(p)) for p in ([3.0, 0.0], [3.1, 1.0], [3.1, -1.0])],
        actuator_list=[ActuatorBCParabolicV(width=width, position_x=0.0, boundary_name=n) for n in ("actuator_up", "actuator_lo")],
    )
    params_ic = flowsolverparameters.ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    return dict(params_flow=params_flow, params_time=params_time, params_save=params_save, params_solver=params_solver,
                params_mesh=params_mesh, params_control=params_control, params_ic=params_ic)


def main(num_steps: int = 1000):
    path_out = Path.cwd() / "data_output"
    fs = CylinderFlowSolver(**make_params(path_out=path_out, num_steps=num_steps, save_every=25), verbose=100)
    fs.compute_steady_state(method="picard", max_iter=3, tol=1e-7, u_ctrl=[0.0, 0.0])
    fs.compute_steady_state(method="newton", max_iter=25, u_ctrl=[0.0, 0.0], initial_guess=fs.fields.UP0)
    fs.initialize_time_stepping(ic=None)
    Kss = Controller.from_file(file=GOLDEN / "controllers" / "Kopt_reduced13.mat", x0=None)
    for _ in range(fs.params_time.num_steps):
        y_meas = flu.MpiUtils.mpi_broadcast(fs.y_meas)
        u_ctrl = Kss.step(y=-y_meas[0], dt=fs.params_time.dt)
        fs.step(u_ctrl=np.repeat(u_ctrl, repeats=2, axis=0))
    fs.write_timeseries()
    flu.summarize_timings(fs)

    fs_restart = CylinderFlowSolver(**make_params(path_out=path_out, num_steps=10, save_every=5, Tstart=25 * 0.005), verbose=5)
    fs_restart.load_steady_state()
    fs_restart.initialize_time_stepping(Tstart=fs_restart.params_time.Tstart)
    for _ in range(fs_restart.params_time.num_steps):
        u_ctrl = Kss.step(y=-fs_restart.y_meas[0], dt=fs_restart.params_time.dt)
        fs_restart.step(u_ctrl=np.repeat(u_ctrl, repeats=2, axis=0))
    fs_restart.write_timeseries()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
