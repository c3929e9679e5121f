"""rocSOLVER pivot-block inverse check (run on the GPU box): which of getrf+getri (in place), getri_outofplace
and getrf+getrs(I) return the right inverse, for every order up to 700 and a sample beyond.
Result on ROCm 7.0 (torch wheel) / MI355X: in-place getri is WRONG for n = 255, 383, 511, 639, 1023, 1151
(n = 127 mod 128); the other two are right everywhere -> fc_refactor uses getri_outofplace."""
import ctypes as C, numpy as np, torch, os, sys, time
torch.zeros(1).cuda()
which = "torch"  # the copies bound to the HIP runtime of this process
d = os.path.dirname(torch.__file__) + "/lib/"
blas = C.CDLL(d + "librocblas.so", mode=C.RTLD_GLOBAL); sol = C.CDLL(d + "librocsolver.so", mode=C.RTLD_GLOBAL)
h = C.c_void_p(); assert blas.rocblas_create_handle(C.byref(h)) == 0
rng = np.random.default_rng(0)
sizes = list(range(1, 700)) + list(range(700, 2100, 7)) + [1023, 1024, 1087, 1151, 1174, 1324, 1416, 2039, 2047, 2048]
bad = {"getri": [], "oop": [], "getrs": [], "getrf": []}
t0 = time.time()
for n in sizes:
    lda = n + 5
    A = rng.standard_normal((n, n)) + n ** 0.5 * np.eye(n)
    Ainv = np.linalg.inv(A); sc = np.abs(Ainv).max()
    buf = np.zeros((n, lda)); buf[:, :n] = A
    ipiv = torch.zeros(n, dtype=torch.int32, device="cuda"); info = torch.zeros(1, dtype=torch.int32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    # 1) getrf + getri in place
    t1 = torch.from_numpy(buf).cuda()
    sol.rocsolver_dgetrf(h, n, n, p(t1), lda, p(ipiv), p(info)); lu = t1.clone()
    sol.rocsolver_dgetri(h, n, p(t1), lda, p(ipiv), p(info)); torch.cuda.synchronize()
    if not np.abs(t1.cpu().numpy()[:, :n] - Ainv).max() <= 1e-9 * sc: bad["getri"].append(n)
    # LU check: buf is row-major A => column-major A^T. P A^T = L U
    # 2) out of place
    t2 = lu.clone(); c2 = torch.zeros((n, lda), dtype=torch.float64, device="cuda")
    sol.rocsolver_dgetri_outofplace(h, n, p(t2), lda, p(ipiv), p(c2), lda, p(info)); torch.cuda.synchronize()
    if not np.abs(c2.cpu().numpy()[:, :n] - Ainv).max() <= 1e-9 * sc: bad["oop"].append(n)
    # 3) getrs with identity
    t3 = lu.clone(); c3 = torch.zeros((n, lda), dtype=torch.float64, device="cuda"); c3[:, :n] = torch.eye(n, dtype=torch.float64, device="cuda")
    sol.rocsolver_dgetrs(h, 111, n, n, p(t3), lda, p(ipiv), p(c3), lda); torch.cuda.synchronize()
    if not np.abs(c3.cpu().numpy()[:, :n] - Ainv).max() <= 1e-9 * sc: bad["getrs"].append(n)
print(which, "tested", len(sizes), "sizes in", round(time.time() - t0, 1), "s")
for k, v in bad.items(): print(k, "bad sizes:", v[:60], len(v))
