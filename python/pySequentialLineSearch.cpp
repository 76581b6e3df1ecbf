// pybind11 module `pySequentialLineSearch`: the two optimizer facades + the three enums, with the snake_case method names
// and keyword arguments of the reference's binding (python/pySequentialLineSearch.cpp:13-153), so the reference's
// python-examples recipes run unchanged on the MI355X path.  Vectors / matrices cross the boundary as numpy float64
// arrays (the reference relies on pybind11/eigen.h; Eigen is not available here, hence the small casters below).
// Additions: set_random_seed(), set/get_global_search_strategy(), set/get_devices(), and batched predict_mean_stdev /
// acquisition_values on (D, M) arrays.
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <sequential-line-search/acquisition-function.hpp>
#include <sequential-line-search/device.hpp>
#include <sequential-line-search/preference-regressor.hpp>
#include <sequential-line-search/preferential-bayesian-optimizer.hpp>
#include <sequential-line-search/sequential-line-search.hpp>
#include <sequential-line-search/utils.hpp>

namespace py = pybind11;
using namespace py::literals;
using namespace sequential_line_search;
using Eigen::MatrixXd;
using Eigen::VectorXd;

#ifndef SLS_HAVE_REAL_EIGEN
namespace pybind11
{
    namespace detail
    {
        template <> struct type_caster<VectorXd>
        {
            PYBIND11_TYPE_CASTER(VectorXd, const_name("numpy.ndarray[float64[n]]"));
            bool load(handle src, bool)
            {
                auto a = array_t<double, array::c_style | array::forcecast>::ensure(src);
                if (!a || a.ndim() != 1) return false;
                value = VectorXd(a.shape(0));
                for (ssize_t i = 0; i < a.shape(0); ++i) value(i) = a.at(i);
                return true;
            }
            static handle cast(const VectorXd& v, return_value_policy, handle)
            {
                array_t<double> a(v.size());
                for (long i = 0; i < v.size(); ++i) a.mutable_at(i) = v(i);
                return a.release();
            }
        };
        template <> struct type_caster<MatrixXd>
        {
            PYBIND11_TYPE_CASTER(MatrixXd, const_name("numpy.ndarray[float64[m, n]]"));
            bool load(handle src, bool)
            {
                auto a = array_t<double, array::forcecast>::ensure(src);
                if (!a || a.ndim() != 2) return false;
                value = MatrixXd(a.shape(0), a.shape(1));
                for (ssize_t i = 0; i < a.shape(0); ++i)
                    for (ssize_t j = 0; j < a.shape(1); ++j) value(i, j) = a.at(i, j);
                return true;
            }
            static handle cast(const MatrixXd& m, return_value_policy, handle)
            {
                array_t<double> a({m.rows(), m.cols()});
                for (long i = 0; i < m.rows(); ++i)
                    for (long j = 0; j < m.cols(); ++j) a.mutable_at(i, j) = m(i, j);
                return a.release();
            }
        };
    } // namespace detail
} // namespace pybind11
#else
#include <pybind11/eigen.h>
#endif

PYBIND11_MODULE(pySequentialLineSearch, m)
{
    m.doc() = "sequential-line-search on AMD MI355X (libsls_hip)";
    m.def("set_random_seed", &utils::SetRandomSeed, "seed"_a);
    // run-time switches of the MI355X build (the reference chooses the maximiser branch at compile time and has one device)
    py::enum_<GlobalSearchStrategy>(m, "GlobalSearchStrategy")
        .value("DirectThenLbfgs", GlobalSearchStrategy::DirectThenLbfgs)
        .value("ParallelMultiStart", GlobalSearchStrategy::ParallelMultiStart);
    m.def("set_global_search_strategy", &acquisition_func::SetGlobalSearchStrategy, "strategy"_a);
    m.def("get_global_search_strategy", &acquisition_func::GetGlobalSearchStrategy);
    m.def("set_local_search_tolerances", &acquisition_func::SetLocalSearchTolerances, "relative_func_tolerance"_a, "relative_param_tolerance"_a);
    m.def("get_local_search_tolerances", [] {
        double f = 0.0, x = 0.0;
        acquisition_func::GetLocalSearchTolerances(&f, &x);
        return std::make_pair(f, x);
    });
    m.def("set_map_fit_tolerances", &acquisition_func::SetMapFitTolerances, "relative_func_tolerance"_a, "relative_param_tolerance"_a);
    m.def("get_map_fit_tolerances", [] {
        double f = 0.0, x = 0.0;
        acquisition_func::GetMapFitTolerances(&f, &x);
        return std::make_pair(f, x);
    });
    m.def("set_devices", &device::SetDevices, "devices"_a);
    m.def("get_devices", [] { return device::Devices(); });

    py::enum_<CurrentBestSelectionStrategy>(m, "CurrentBestSelectionStrategy", py::arithmetic())
        .value("LargestExpectValue", CurrentBestSelectionStrategy::LargestExpectValue)
        .value("LastSelection", CurrentBestSelectionStrategy::LastSelection);
    py::enum_<AcquisitionFuncType>(m, "AcquisitionFuncType", py::arithmetic())
        .value("ExpectedImprovement", AcquisitionFuncType::ExpectedImprovement)
        .value("GaussianProcessUpperConfidenceBound", AcquisitionFuncType::GaussianProcessUpperConfidenceBound);
    py::enum_<KernelType>(m, "KernelType", py::arithmetic())
        .value("ArdSquaredExponentialKernel", KernelType::ArdSquaredExponentialKernel)
        .value("ArdMatern52Kernel", KernelType::ArdMatern52Kernel);

    using SliderEndsGenerator = std::function<std::pair<VectorXd, VectorXd>(const int)>;
    m.def("generate_random_slider_ends", &GenerateRandomSliderEnds, "num_dims"_a);
    m.def("generate_centered_fixed_length_random_slider_ends", &GenerateCenteredFixedLengthRandomSliderEnds, "num_dims"_a);

    py::class_<SequentialLineSearchOptimizer>(m, "SequentialLineSearchOptimizer")
        .def(py::init<const int, const bool, const bool, const KernelType, const AcquisitionFuncType, const SliderEndsGenerator&,
                      const CurrentBestSelectionStrategy>(),
             "num_dims"_a, "use_slider_enlargement"_a = true, "use_map_hyperparams"_a = true,
             "kernel_type"_a = KernelType::ArdMatern52Kernel, "acquisition_func_type"_a = AcquisitionFuncType::ExpectedImprovement,
             "initial_query_generator"_a         = SliderEndsGenerator(GenerateRandomSliderEnds),
             "current_best_selection_strategy"_a = CurrentBestSelectionStrategy::LargestExpectValue)
        .def("set_hyperparams", &SequentialLineSearchOptimizer::SetHyperparams, "kernel_signal_var"_a = 0.500,
             "kernel_length_scale"_a = 0.500, "noise_level"_a = 0.005, "kernel_hyperparams_prior_var"_a = 0.250, "btl_scale"_a = 0.010)
        .def("submit_feedback_data", static_cast<void (SequentialLineSearchOptimizer::*)(const double)>(&SequentialLineSearchOptimizer::SubmitFeedbackData),
             "slider_position"_a)
        .def("submit_feedback_data",
             static_cast<void (SequentialLineSearchOptimizer::*)(const double, const int, const int, const int)>(
                 &SequentialLineSearchOptimizer::SubmitFeedbackData),
             "slider_position"_a, "num_map_estimation_iters"_a, "num_global_search_iters"_a, "num_local_search_iters"_a)
        .def("get_slider_ends", &SequentialLineSearchOptimizer::GetSliderEnds)
        .def("calc_point_from_slider_position", &SequentialLineSearchOptimizer::CalcPointFromSliderPosition, "slider_position"_a)
        .def("get_maximizer", &SequentialLineSearchOptimizer::GetMaximizer)
        .def("get_preference_value_mean", &SequentialLineSearchOptimizer::GetPreferenceValueMean, "point"_a)
        .def("get_preference_value_stdev", &SequentialLineSearchOptimizer::GetPreferenceValueStdev, "point"_a)
        .def("get_acquisition_func_value", &SequentialLineSearchOptimizer::GetAcquisitionFuncValue, "point"_a)
        .def("get_preference_value_means", &SequentialLineSearchOptimizer::GetPreferenceValueMeans, "points"_a)
        .def("get_preference_value_stdevs", &SequentialLineSearchOptimizer::GetPreferenceValueStdevs, "points"_a)
        .def("get_acquisition_func_values", &SequentialLineSearchOptimizer::GetAcquisitionFuncValues, "points"_a)
        .def("get_raw_data_points", &SequentialLineSearchOptimizer::GetRawDataPoints)
        .def("damp_data", &SequentialLineSearchOptimizer::DampData, "directory_path"_a)
        .def("set_gaussian_process_upper_confidence_bound_hyperparam",
             &SequentialLineSearchOptimizer::SetGaussianProcessUpperConfidenceBoundHyperparam, "hyperparam"_a);

    py::class_<PreferentialBayesianOptimizer>(m, "PreferentialBayesianOptimizer")
        .def(py::init<const int, const bool, const KernelType, const AcquisitionFuncType, const InitialQueryGenerator&,
                      const CurrentBestSelectionStrategy, const int>(),
             "num_dims"_a, "use_map_hyperparams"_a = true, "kernel_type"_a = KernelType::ArdMatern52Kernel,
             "acquisition_func_type"_a = AcquisitionFuncType::ExpectedImprovement,
             "initial_query_generator"_a         = InitialQueryGenerator(GenerateRandomPoints),
             "current_best_selection_strategy"_a = CurrentBestSelectionStrategy::LargestExpectValue, "num_options"_a = 2)
        .def("set_hyperparams", &PreferentialBayesianOptimizer::SetHyperparams, "kernel_signal_var"_a = 0.500,
             "kernel_length_scale"_a = 0.500, "noise_level"_a = 0.005, "kernel_hyperparams_prior_var"_a = 0.250, "btl_scale"_a = 0.010)
        .def("submit_feedback_data", &PreferentialBayesianOptimizer::SubmitFeedbackData, "option_index"_a, "num_map_estimation_iters"_a = 0)
        .def("submit_custom_feedback_data", &PreferentialBayesianOptimizer::SubmitCustomFeedbackData, "chosen_option"_a,
             "other_options"_a, "num_map_estimation_iters"_a = 0)
        .def("determine_next_query", &PreferentialBayesianOptimizer::DetermineNextQuery, "num_global_search_iters"_a = 0,
             "num_local_search_iters"_a = 0)
        .def("get_current_options", &PreferentialBayesianOptimizer::GetCurrentOptions)
        .def("get_maximizer", &PreferentialBayesianOptimizer::GetMaximizer)
        .def("get_preference_value_mean", &PreferentialBayesianOptimizer::GetPreferenceValueMean, "point"_a)
        .def("get_preference_value_stdev", &PreferentialBayesianOptimizer::GetPreferenceValueStdev, "point"_a)
        .def("get_acquisition_func_value", &PreferentialBayesianOptimizer::GetAcquisitionFuncValue, "point"_a)
        .def("get_preference_value_means", &PreferentialBayesianOptimizer::GetPreferenceValueMeans, "points"_a)
        .def("get_preference_value_stdevs", &PreferentialBayesianOptimizer::GetPreferenceValueStdevs, "points"_a)
        .def("get_acquisition_func_values", &PreferentialBayesianOptimizer::GetAcquisitionFuncValues, "points"_a)
        .def("get_raw_data_points", &PreferentialBayesianOptimizer::GetRawDataPoints)
        .def("damp_data", &PreferentialBayesianOptimizer::DampData, "directory_path"_a)
        .def("set_gaussian_process_upper_confidence_bound_hyperparam",
             &PreferentialBayesianOptimizer::SetGaussianProcessUpperConfidenceBoundHyperparam, "hyperparam"_a);
}
