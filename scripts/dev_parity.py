import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from cmdg_loader import cm
from oracle import oracle as O
from helpers import pseudo1d_setup, rel_linf
for direction in (0, 1, 2):
  for fbc in (False, True):
    law, grid, dt = pseudo1d_setup(direction=direction, flux_bc=fbc)
    odg = O.OracleDGModel(law, grid, nf_first=0, direction=direction)
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
    dg = cm.dgmodel.DGModel(law, grid, direction=direction)
    Q = dg.init_ode_state(0.0)
    assert np.array_equal(Q.cpu().numpy(), Q0)
    # raw tendency parity (alpha, beta variants)
    rng = np.random.default_rng(1)
    T0 = rng.standard_normal(Q0.shape)
    for (al, be) in ((1.0, 0.0), (1.0, 1.0), (0.5, 2.0)):
        To = T0.copy(); odg(To, Q0.copy(), 0.3, al, be)
        Tg = torch.from_numpy(T0.copy()).cuda(); torch.cuda.synchronize()
        dg(Tg, Q, 0.3, al, be)
        Tg = Tg.cpu().numpy()
        print("dir", direction, "fbc", fbc, "ab", al, be, "tendency rel Linf", rel_linf(Tg[:grid.nreal], To[:grid.nreal]),
              "gf", rel_linf(dg.state_gradient_flux.cpu().numpy()[:grid.nreal], odg.state_gradient_flux[:grid.nreal]))
    # full run to t=1
    solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt, t0=0.0)
    t0 = time.time()
    cm.odesolvers.solve(Q, solver, timeend=1.0)
    Qe = dg.init_ode_state(1.0)
    err = dg.euclidean_distance(Q, Qe)
    exp = {0:9.6252415559793265e-03,1:1.7475667486259477e-02,2:6.6162204724938736e-02}[direction]
    Qo = Q0.copy(); O.solve(odg, Qo, dt, 1.0)
    print("  L2 err %.16e golden %.16e rel %.2e ; vs oracle state rel Linf %.2e (%.2fs)" % (err, exp, abs(err-exp)/exp, rel_linf(Q.cpu().numpy()[:grid.nreal], Qo[:grid.nreal]), time.time()-t0))
    dg.close()
