"""Oracle run behind tests/golden/cavity_coarse_re500.npz: cavity_coarse, Re=500, Picard(10, tol 1e-7) → Newton(10),
Frobenius norm of the steady Jacobian A = -dF/dUP0 with bc.apply (reference test_operatorgetter.py:23-26,133-144)."""
import sys, tempfile, numpy as np, time
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
from tests.support import ndsolver
from flowcontrol_amd.fem.boundary import combine_bcs
from flowcontrol_amd.examples.cavity.cavityflowsolver import CavityFlowSolver
from oracle import ns_oracle as O
fs=CavityFlowSolver.make_default(Re=500, path_out=tempfile.mkdtemp()); th=fs.th; d=O.Disc.from_taylor_hood(th)
dofs_full, vals_full = combine_bcs(fs._make_BCs().bcu, th.N)
skip=np.zeros(th.N,bool); skip[dofs_full]=True
perm=ndsolver.build_tree(th.cell_dofs, th.mesh.cell_centroids(), th.N, 13, skip).perm
up=np.zeros(th.N); g=fs._default_steady_state_initial_guess()(th.node_coords); up[:th.nn]=g[:,0]
up=O.picard(d,1/500.,up,dofs_full,vals_full,max_iter=10,tol=1e-7,perm=perm,log=print)
up=O.newton(d,1/500.,up,dofs_full,vals_full,max_iter=10,perm=perm,log=print)
pd,_=combine_bcs(fs.bc.bcu, th.N)
A=O.steady_jacobian_A(d,1/500.,up,pd)
fro=np.sqrt((A.data**2).sum())
print('FROB %.14g ref 47.31849925281407'%fro)
np.savez_compressed(str(__import__('pathlib').Path(__file__).parent / 'cavity_coarse_re500.npz'), frob=fro, u0max=up[:2*th.nn].max(), u0mean=up[:2*th.nn].mean())
