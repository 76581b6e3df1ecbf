"""The recipes of the reference's python-examples through the pybind11 module `pySequentialLineSearch`, UNCHANGED: the same
constructor keywords (none the scripts do not pass -- the default constructor estimates the hyper-parameters jointly,
use_map_hyperparams = True), the same set_hyperparams values, 30 iterations, the same simulated user (a 1000-sample scan of the
slider / the better of two options).  Restated here, not copied: what is pinned is the API surface and the behaviour.

Reference recipes: python-examples/simple.py:32-47, custom-initial-slider.py:38-57, pairwise-comparison-query.py:39-65, and the
constructor forms of kernel-comparison.py:111-119, acquisition-func-comparison.py:93-101, map-vs-fixed-hyperparams.py:102-112.
The scripts print a residual per iteration and assert nothing; asserted here: every call succeeds, the maximiser stays within a
minimum slider length of the unit cube, and the residual to the optimum 0.2 * 1 ends well below where it started (the simulated user answers exactly)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pysls():
    sys.path.insert(0, os.path.join(ROOT, "sequential-line-search_amd"))
    import pySequentialLineSearch
    return pySequentialLineSearch


def simulated_objective(x):
    return -np.linalg.norm(x - 0.2)


def simulated_slider_user(slider_ends):
    """The scripts' stand-in for the human: the best of 1000 evenly spaced slider positions, first maximum."""
    ts = np.arange(1000) / 999.0
    f = [simulated_objective((1.0 - t) * slider_ends[0] + t * slider_ends[1]) for t in ts]
    return float(ts[int(np.argmax(f))])


def random_initial_slider(num_dims):
    return np.random.uniform(0.0, 1.0, (num_dims,)), np.random.uniform(0.0, 1.0, (num_dims,))


def run_line_search(optimizer, iters=30):
    res = []
    for _ in range(iters):
        optimizer.submit_feedback_data(simulated_slider_user(optimizer.get_slider_ends()))
        x = optimizer.get_maximizer()
        # a slider shorter than its minimum length (0.25) is stretched WITHOUT being cropped to the unit cube (src/slider.cpp:104-118):
        # the points a user picks on it, and with them the maximiser, may leave the cube by up to that length
        assert x.shape == (5,) and np.all(x >= -0.25) and np.all(x <= 1.25)
        res.append(float(np.linalg.norm(x - 0.2)))
    return res


def test_simple_recipe(pysls):
    """simple.py: default constructor (MAP hyper-parameters on), 30 iterations."""
    pysls.set_random_seed(3)                         # addition of this build: the reference draws from an unseeded std::rand
    optimizer = pysls.SequentialLineSearchOptimizer(num_dims=5)
    optimizer.set_hyperparams(kernel_signal_var=0.50, kernel_length_scale=0.10, kernel_hyperparams_prior_var=0.10)
    res = run_line_search(optimizer)
    assert len(res) == 30 and res[-1] < 0.5 * res[0] and res[-1] < 0.2, res
    assert optimizer.get_raw_data_points().shape[0] == 5
    assert optimizer.get_preference_value_stdev(optimizer.get_maximizer()) >= 0.0


def test_custom_initial_slider_recipe(pysls):
    """custom-initial-slider.py: use_map_hyperparams=True + a Python callable as the first slider's generator."""
    pysls.set_random_seed(4)
    np.random.seed(4)
    optimizer = pysls.SequentialLineSearchOptimizer(num_dims=5, use_map_hyperparams=True, initial_query_generator=random_initial_slider)
    optimizer.set_hyperparams(kernel_signal_var=0.50, kernel_length_scale=0.10, kernel_hyperparams_prior_var=0.10)
    res = run_line_search(optimizer)
    assert res[-1] < 0.5 * res[0] and res[-1] < 0.2, res


def test_pairwise_comparison_recipe(pysls):
    """pairwise-comparison-query.py: PreferentialBayesianOptimizer with its defaults (MAP on, two options), LastSelection, a Python
    initial-query generator; 30 x (submit_feedback_data, determine_next_query()) with no explicit iteration counts."""
    pysls.set_random_seed(5)
    np.random.seed(5)
    optimizer = pysls.PreferentialBayesianOptimizer(num_dims=5, initial_query_generator=lambda nd, no: [np.random.rand(nd) for _ in range(no)],
                                                    current_best_selection_strategy=pysls.CurrentBestSelectionStrategy.LastSelection)
    optimizer.set_hyperparams(kernel_signal_var=0.50, kernel_length_scale=0.10, kernel_hyperparams_prior_var=0.10)
    res = []
    for _ in range(30):
        options = optimizer.get_current_options()
        assert len(options) == 2
        optimizer.submit_feedback_data(int(np.argmax([simulated_objective(x) for x in options])))
        optimizer.determine_next_query()
        res.append(float(np.linalg.norm(optimizer.get_maximizer() - 0.2)))
    # pairwise comparisons carry far less information per query than a line search: the trend is what the script shows
    assert np.all(np.isfinite(res)) and min(res[-10:]) < res[0], res


@pytest.mark.parametrize("form", ["kernel_se", "kernel_matern", "ucb", "fixed_hyperparams"])
def test_comparison_script_constructor_forms(pysls, form):
    """The constructor keyword sets of the three comparison scripts (their sweeps and plots are not part of the path): every
    kernel type, GP-UCB with set_gaussian_process_upper_confidence_bound_hyperparam, and use_map_hyperparams False / True."""
    pysls.set_random_seed(6)
    np.random.seed(6)
    kw = dict(num_dims=5, use_map_hyperparams=True, initial_query_generator=random_initial_slider)
    if form == "kernel_se":
        kw["kernel_type"] = pysls.KernelType.ArdSquaredExponentialKernel
    elif form == "kernel_matern":
        kw["kernel_type"] = pysls.KernelType.ArdMatern52Kernel
    elif form == "ucb":
        kw["acquisition_func_type"] = pysls.AcquisitionFuncType.GaussianProcessUpperConfidenceBound
    else:
        kw["use_map_hyperparams"] = False
        kw["kernel_type"] = pysls.KernelType.ArdMatern52Kernel
    optimizer = pysls.SequentialLineSearchOptimizer(**kw)
    if form == "ucb":
        optimizer.set_gaussian_process_upper_confidence_bound_hyperparam(1.0)
    else:
        optimizer.set_hyperparams(kernel_signal_var=0.50, kernel_length_scale=0.25, kernel_hyperparams_prior_var=0.10)
    res = run_line_search(optimizer, iters=12)
    assert res[-1] < res[0], (form, res)


def test_build_specific_switches(pysls):
    """Additions of this build (not in the reference's module): maximiser branch and device list at run time."""
    assert pysls.get_devices() == [0]
    before = pysls.get_global_search_strategy()
    pysls.set_global_search_strategy(pysls.GlobalSearchStrategy.ParallelMultiStart)
    assert pysls.get_global_search_strategy() == pysls.GlobalSearchStrategy.ParallelMultiStart
    pbo = pysls.PreferentialBayesianOptimizer(num_dims=2, num_options=3)
    for _ in range(3):
        o = pbo.get_current_options()
        pbo.submit_feedback_data(int(np.argmax([simulated_objective(x) for x in o])))
        pbo.determine_next_query(32, 10)
    assert len(pbo.get_current_options()) == 3
    pysls.set_global_search_strategy(before)


def test_tolerance_setters_of_both_kinds_of_search(pysls):
    """The local searches start with nloptutil::solve's 1e-6 / 1e-6, the MAP fits with 0 / 0 (off: the documented deviation,
    INTEGRATION.md 2); both pairs have run-time setters (round 6 added the MAP fits': ADVICE), and a line search with the MAP fits
    under the reference's tolerances still improves."""
    assert pysls.get_local_search_tolerances() == (1e-6, 1e-6)
    assert pysls.get_map_fit_tolerances() == (0.0, 0.0)
    pysls.set_map_fit_tolerances(1e-6, 1e-6)
    try:
        assert pysls.get_map_fit_tolerances() == (1e-6, 1e-6)
        optimizer = pysls.SequentialLineSearchOptimizer(num_dims=5)
        res = run_line_search(optimizer, iters=8)
        assert res[-1] < res[0], res
    finally:
        pysls.set_map_fit_tolerances(0.0, 0.0)
    assert pysls.get_map_fit_tolerances() == (0.0, 0.0)


def test_batched_accessors_match_the_scalar_ones(pysls):
    """batched mean / deviation / acquisition value over the columns of a D x M matrix (one device pass each;
    the reference's GUI demo calls the scalar forms pixel by pixel, demos/bayesian_optimization_2d_gui/mainwidget.cpp:41-72).
    Same numbers as the scalar accessors, point by point, for both optimisers; zeros before any data."""
    np.random.seed(3)
    opt = pysls.SequentialLineSearchOptimizer(num_dims=5)
    P = np.asfortranarray(np.random.uniform(0, 1, (5, 33)))
    assert np.all(opt.get_preference_value_means(P) == 0.0) and np.all(opt.get_acquisition_func_values(P) == 0.0)
    for _ in range(4):
        opt.submit_feedback_data(simulated_slider_user(opt.get_slider_ends()))
    pbo = pysls.PreferentialBayesianOptimizer(num_dims=5, num_options=3)
    for _ in range(3):
        o = pbo.get_current_options()
        pbo.submit_feedback_data(int(np.argmax([simulated_objective(x) for x in o])))
        pbo.determine_next_query(32, 10)
    for obj in (opt, pbo):
        mu, sg, av = obj.get_preference_value_means(P), obj.get_preference_value_stdevs(P), obj.get_acquisition_func_values(P)
        assert mu.shape == sg.shape == av.shape == (33,)
        for m in range(33):
            x = P[:, m].copy()
            assert mu[m] == pytest.approx(obj.get_preference_value_mean(x), rel=1e-9, abs=1e-12)
            assert sg[m] == pytest.approx(obj.get_preference_value_stdev(x), rel=1e-7, abs=1e-10)
            assert av[m] == pytest.approx(obj.get_acquisition_func_value(x), rel=1e-7, abs=1e-12)
