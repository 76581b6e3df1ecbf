// Small helpers (reference: include/sequential-line-search/utils.hpp): random vectors, Bradley-Terry-Luce model, CSV export.
#ifndef SEQUENTIAL_LINE_SEARCH_UTILS_HPP
#define SEQUENTIAL_LINE_SEARCH_UTILS_HPP

#include <cmath>
#include <sequential-line-search/eigen-lite.hpp>
#include <string>
#include <vector>

namespace sequential_line_search
{
    namespace utils
    {
        /// Uniform sample from [0, 1]^n.
        Eigen::VectorXd GenerateRandomVector(unsigned n);

        /// Seeds the stream behind GenerateRandomVector / the multi-start start sets (the reference never seeds
        /// std::rand; an explicit seed makes runs reproducible -- SURVEY.md Appendix B.3).
        void SetRandomSeed(unsigned long long seed);

        /// BTL probability that f(0) is chosen among f: exp(f0/s) / sum_j exp(fj/s)  (no max-subtraction, as the reference).
        inline double CalcBtl(const Eigen::VectorXd& f, double scale = 1.0)
        {
            double sum = 0.0;
            for (long i = 0; i < f.size(); ++i) sum += std::exp(f(i) / scale);
            return std::exp(f(0) / scale) / sum;
        }

        inline Eigen::VectorXd CalcBtlDerivative(const Eigen::VectorXd& f, double scale = 1.0)
        {
            const double    btl = CalcBtl(f, scale);
            const double    c   = -btl * btl / scale;
            Eigen::VectorXd d(f.size());
            double          sum = 0.0;
            for (long i = 1; i < f.size(); ++i)
            {
                const double e = std::exp((f(i) - f(0)) / scale);
                d(i)           = c * e;
                sum += e;
            }
            d(0) = -c * sum;
            return d;
        }

        void ExportMatrixToCsv(const std::string& file_path, const Eigen::MatrixXd& X);

        /// Reads a matrix written by ExportMatrixToCsv (comma separated, one matrix row per line).  The reference only
        /// writes its state (DampData); this loader makes the dump usable for resuming a session.
        Eigen::MatrixXd ImportMatrixFromCsv(const std::string& file_path);

        /// Reads the preference tuples of DampData's D.csv (one comma separated index tuple per line).
        std::vector<std::vector<unsigned>> ImportPreferencesFromCsv(const std::string& file_path);
    } // namespace utils
} // namespace sequential_line_search

#endif
