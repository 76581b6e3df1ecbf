"""Host-side logic that needs no GPU: the scheduling model used to design the dynamic pools of the fused factor + inverse launch."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scheduler_model_of_the_fused_inverse_runs():
    """tools/potri_sched_sim.py (the discrete-event model that preceded the dynamic pools of kernels_chol.hip): static ownership,
    one pool, one pool per XCD on a small problem -- every item finishes, pooling never loses by more than the claim's cost, and the
    chain cannot end before nb diagonal blocks have run."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("potri_sched_sim", os.path.join(ROOT, "tools", "potri_sched_sim.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    nb = 10
    t_static, c_static = m.simulate(nb, 24, 40, 0, verbose=False)
    t_pool, c_pool = m.simulate(nb, 24, 40, 1, verbose=False)
    t_xcd, c_xcd = m.simulate(nb, 24, 40, 4, verbose=False)
    for t, c in ((t_static, c_static), (t_pool, c_pool), (t_xcd, c_xcd)):
        assert c >= nb * 19.5 and t >= c
    assert t_pool <= t_static * 1.05 and t_xcd <= t_static * 1.10
