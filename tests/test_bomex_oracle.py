"""BOMEX pieces of the moist LES law in the oracle (experiments/AtmosLES/bomex_model.jl): the
reference stores no number for this experiment and its thermodynamics package is not in the tree,
so parity with the reference is UNPINNED here; what is checked is that the restated sources and
surface conditions do what their definitions say: budgets of mass and total water close against
the prescribed surface flux and the volume sources, the sources match a direct evaluation of
their formulas, the drag law opposes the near-surface wind.  CPU only."""
import numpy as np

from helpers import bomex_setup


def test_bomex_budgets_close(oracle):
    law, grid = bomex_setup(nx=3, ny=3, nz=8)
    dg = oracle.OracleDGModel(law, grid)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    rng = np.random.default_rng(1)
    Q[:, 1:4] += Q[:, 0:1] * 0.5 * rng.standard_normal(Q[:, 1:4].shape)
    T = np.zeros_like(Q)
    dg(T, Q, 0.0, 1.0, 0.0)
    assert np.isfinite(T).all()
    M = grid.vgeo[:grid.nreal, 9, :]
    # the same evaluation without sources and with the default (no-flux) surface
    law0, _ = bomex_setup(nx=3, ny=3, nz=8)
    law0.sources = 1
    law0.boundary_conditions = (1, 1)
    dg0 = oracle.OracleDGModel(law0, grid)
    T0 = np.zeros_like(Q)
    dg0(T0, Q.copy(), 0.0, 1.0, 0.0)
    area = 1200.0 * 1200.0
    b = law.bomex
    for s in (0, 5):      # rho and rho q_tot receive the same surface flux and volume source
        closed = (M * T0[:grid.nreal, s]).sum()
        assert abs(closed) <= 1e-9 * (M * np.abs(T0[:grid.nreal, s])).sum()
        # volume source of BomexTendencies, evaluated from its definition
        z = dg.state_auxiliary[:grid.nreal, 3] / law.ps.grav
        rho = Q[:grid.nreal, 0]
        lin = (z - b["zl_moisture"]) / (b["zh_moisture"] - b["zl_moisture"])
        rdqt = np.where(z <= b["zl_moisture"], rho * b["dqt_peak"],
                        np.where(z <= b["zh_moisture"], rho * (b["dqt_peak"] - b["dqt_peak"] * lin), 0.0))
        ls = (z - b["zl_sub"]) / (b["zh_sub"] - b["zl_sub"])
        w_s = np.where(z <= b["zl_sub"], z * b["w_sub"] / b["zl_sub"],
                       np.where(z <= b["zh_sub"], b["w_sub"] - b["w_sub"] * ls, 0.0))
        dqdz = dg.state_gradient_flux[:grid.nreal, 3 + 7 + 2]
        vol = (M * (rdqt - rho * w_s * dqdz)).sum()
        total = (M * T[:grid.nreal, s]).sum()
        assert abs(total - (vol + b["q_flux"] * area)) <= 1e-9 * abs(b["q_flux"] * area)


def test_bomex_sources_match_their_definitions(oracle):
    """momentum sources at rest relative to the geostrophic wind vanish; the sponge acts only
    above z_sponge; the Coriolis term turns the ageostrophic wind to the right."""
    law, grid = bomex_setup(nx=2, ny=2, nz=10)
    b, ps = law.bomex, law.ps
    dg = oracle.OracleDGModel(law, grid)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    z = dg.state_auxiliary[:, 3] / ps.grav
    # put the flow exactly on the geostrophic profile
    Q[:, 1] = Q[:, 0] * (b["u_geostrophic"] + b["u_slope"] * z)
    Q[:, 2] = 0.0
    Q[:, 3] = 0.0
    law_g, _ = bomex_setup(nx=2, ny=2, nz=10)
    law_g.sources = 4 | 8                       # sponge + geostrophic only
    law_g.boundary_conditions = (1, 1)
    law_n, _ = bomex_setup(nx=2, ny=2, nz=10)
    law_n.sources = 0
    law_n.boundary_conditions = (1, 1)
    Tg, Tn = np.zeros_like(Q), np.zeros_like(Q)
    oracle.OracleDGModel(law_g, grid)(Tg, Q.copy(), 0.0, 1.0, 0.0)
    oracle.OracleDGModel(law_n, grid)(Tn, Q.copy(), 0.0, 1.0, 0.0)
    assert np.abs(Tg[:, 1:4] - Tn[:, 1:4]).max() < 1e-14
    # an ageostrophic zonal wind du: Coriolis gives d(rho v)/dt = -f rho du, nothing in x; the
    # sponge damps it above z_sponge only
    du = 2.0
    Q[:, 1] += Q[:, 0] * du
    oracle.OracleDGModel(law_g, grid)(Tg, Q.copy(), 0.0, 1.0, 0.0)
    oracle.OracleDGModel(law_n, grid)(Tn, Q.copy(), 0.0, 1.0, 0.0)
    S = Tg - Tn
    nr = grid.nreal
    low = z[:nr] < b["z_sponge"] - 1e-9
    assert np.abs(S[:nr, 1][low]).max() < 1e-13
    assert np.allclose(S[:nr, 2][low], (-b["f_coriolis"] * Q[:nr, 0] * du)[low], rtol=1e-12)
    high = z[:nr] > b["z_sponge"] + 50.0
    r = (z[:nr] - b["z_sponge"]) / (b["z_max"] - b["z_sponge"])
    beta = b["alpha_max"] * np.sin(np.pi * r / 2) ** 2
    assert np.allclose(S[:nr, 1][high], (-beta * Q[:nr, 0] * du)[high], rtol=1e-12)


def test_bomex_surface_drag_opposes_the_wind(oracle):
    law, grid = bomex_setup(nx=2, ny=2, nz=6)
    law.sources = 1
    dg = oracle.OracleDGModel(law, grid)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    law_n, _ = bomex_setup(nx=2, ny=2, nz=6)
    law_n.sources = 1
    law_n.boundary_conditions = (1, 1)
    T, Tn = np.zeros_like(Q), np.zeros_like(Q)
    dg(T, Q.copy(), 0.0, 1.0, 0.0)
    oracle.OracleDGModel(law_n, grid)(Tn, Q.copy(), 0.0, 1.0, 0.0)
    M = grid.vgeo[:grid.nreal, 9, :]
    drag = (M * (T[:grid.nreal, 1] - Tn[:grid.nreal, 1])).sum()      # wind is -8.75 m/s in x
    b = law.bomex
    rho_sfc = Q[:grid.nreal, 0].max()
    area = 800.0 * 800.0
    # the moisture flux carries momentum rho u too: q_flux * u; the stress is rho u_star^2
    expect = rho_sfc * b["u_star"] ** 2 * area + b["q_flux"] * (-8.75) * area
    assert drag > 0 and abs(drag - expect) < 2e-2 * abs(expect)
