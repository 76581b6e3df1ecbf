#include <cstdlib>
#include <fstream>
#include <sequential-line-search/utils.hpp>

namespace sequential_line_search
{
    namespace utils
    {
        // reference: src/utils.cpp:8-11
        Eigen::VectorXd GenerateRandomVector(unsigned n) { return 0.5 * (Eigen::VectorXd::Random(n) + Eigen::VectorXd::Ones(n)); }

        void SetRandomSeed(unsigned long long seed)
        {
#ifndef SLS_HAVE_REAL_EIGEN
            Eigen::lite::RandomState() = seed;
#else
            std::srand(static_cast<unsigned>(seed));
#endif
        }

        // reference: src/utils.cpp:13-18 (comma separated, one matrix row per line)
        void ExportMatrixToCsv(const std::string& file_path, const Eigen::MatrixXd& X)
        {
            std::ofstream file(file_path);
            for (long i = 0; i < X.rows(); ++i)
            {
                for (long j = 0; j < X.cols(); ++j) file << X(i, j) << (j + 1 != X.cols() ? "," : "");
                if (i + 1 != X.rows()) file << "\n";
            }
        }
    } // namespace utils
} // namespace sequential_line_search
