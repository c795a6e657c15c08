"""The preconditions ``cmdg_create`` puts on the caller's grid tables (include/cmdg.h) and the
lifetime rule of nested handles.

The kernels compute ``vmap-`` from (face, node) and read ``vMI`` from ``vgeo`` instead of loading
the reference's tables, so create verifies on the device that the caller's tables say the same and
refuses a grid for which they do not -- loudly, with the reason, never by computing something
else than the reference would.
"""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _grid():
    from helpers import pseudo1d_setup
    law, grid, _ = pseudo1d_setup(Ne=2)
    return law, grid


def test_create_rejects_a_reordered_vmapM(cm, torch):
    law, grid = _grid()
    g = copy.copy(grid)
    g.vmapM = grid.vmapM.copy()
    a, b = g.vmapM[1, 2, 0], g.vmapM[1, 2, 1]
    g.vmapM[1, 2, 0], g.vmapM[1, 2, 1] = b, a          # two face nodes swapped: still a valid map
    with pytest.raises(cm._lib.CmdgError, match="vmapM is not the face numbering"):
        cm.dgmodel.DGModel(law, g)


def test_create_rejects_sgeo_whose_vMI_is_not_vgeo_MI(cm, torch):
    law, grid = _grid()
    g = copy.copy(grid)
    g.sgeo = grid.sgeo.copy()
    g.sgeo[0, 0, 0, 4] = np.nextafter(g.sgeo[0, 0, 0, 4], np.inf)     # one ulp off
    with pytest.raises(cm._lib.CmdgError, match="vMI differs from vgeo's MI"):
        cm.dgmodel.DGModel(law, g)


def test_create_accepts_the_unmodified_grid_and_reports_no_neighbours(cm, torch):
    law, grid = _grid()
    dg = cm.dgmodel.DGModel(law, grid)
    assert dg.query("DIRECT_SEND") == 0 and dg.query("DIRECT_RECV") == 0 and dg.query("HALO_PIPELINE") == 0
    assert dg.query(("AUX_READ", 3)) == law.naux          # a law that declares nothing: every column
    dg.close()


def test_nested_handle_may_be_destroyed_first(cm, torch):
    """``hooks.pre_rhs_handle``: destroying the nested Continuity3d operator before its parent
    detaches it (include/cmdg.h); the parent then refuses to evaluate -- it would compute another
    law than the one it was given -- instead of reading freed memory or going on silently, until
    new hooks are set.  (OceanDGModel01.close() still releases them in the safe order.)"""
    from helpers import simple_box_2dt_setup
    model, g3, _, _ = simple_box_2dt_setup(Nx=3, Ny=3, Nz=3)
    odg = cm.ocean01.OceanDGModel01(model, g3)
    Q = odg.dg.init_ode_state(0.0)
    T1, T2 = odg.dg.create_state(), odg.dg.create_state()
    torch.cuda.synchronize()
    odg.dg(T1, Q, 0.0, 1.0, 0.0)
    odg.conti3d_dg.close()                      # the child goes first
    with pytest.raises(RuntimeError, match="nested operator of this handle was destroyed"):
        odg.dg(T2, Q, 0.0, 1.0, 0.0)            # no use after free, and no other physics either
    odg.dg.set_rhs_hooks()
    odg.dg(T2, Q, 0.0, 1.0, 0.0)                # without hooks: the plain operator again
    assert torch.isfinite(T2[:g3.nreal]).all()
    for f in (odg.fu, odg.ft):
        f.close()
    odg.dg.close()


def test_gradient_flux_copies_are_refused_on_a_node_major_law(cm, torch):
    """``cmdg_set_rhs_hooks`` with gradient-flux -> auxiliary copies addresses ``state_gradient_flux``
    in the reference layout; the dry atmosphere keeps it node-major (cmdg.h,
    ``cmdg_export_gradient_flux``) and says so instead of copying the wrong numbers."""
    from helpers import rising_bubble_setup
    law, grid = rising_bubble_setup(nx=2, ny=2, nz=2)
    dg = cm.dgmodel.DGModel(law, grid)
    with pytest.raises(RuntimeError, match="node-major"):
        dg.set_rhs_hooks(gradflux_to_aux=[(0, law.naux - 1, 1.0)])
    dg.close()
