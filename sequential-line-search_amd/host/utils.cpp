#include <stdexcept>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <string>
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
            file << std::setprecision(17);
            for (long i = 0; i < X.rows(); ++i)
            {
                for (long j = 0; j < X.cols(); ++j) file << X(i, j) << (j + 1 != X.cols() ? "," : "");
                if (i + 1 != X.rows()) file << "\n";
            }
        }
        static std::vector<std::vector<double>> ReadCsvRows(const std::string& file_path)
        {
            std::ifstream                    file(file_path);
            std::vector<std::vector<double>> rows;
            std::string                      line;
            while (std::getline(file, line))
            {
                if (line.empty()) continue;
                std::vector<double> row;
                size_t              pos = 0;
                while (pos <= line.size())
                {
                    const size_t next = line.find(',', pos);
                    row.push_back(std::stod(line.substr(pos, next == std::string::npos ? std::string::npos : next - pos)));
                    if (next == std::string::npos) break;
                    pos = next + 1;
                }
                rows.push_back(row);
            }
            return rows;
        }

        Eigen::MatrixXd ImportMatrixFromCsv(const std::string& file_path)
        {
            const auto      rows = ReadCsvRows(file_path);
            Eigen::MatrixXd X(static_cast<long>(rows.size()), rows.empty() ? 0 : static_cast<long>(rows[0].size()));
            for (size_t i = 0; i < rows.size(); ++i)
            {
                if (rows[i].size() != rows[0].size())
                    throw std::runtime_error("ImportMatrixFromCsv: ragged row " + std::to_string(i) + " in " + file_path);
                for (size_t j = 0; j < rows[i].size(); ++j) X(static_cast<long>(i), static_cast<long>(j)) = rows[i][j];
            }
            return X;
        }

        std::vector<std::vector<unsigned>> ImportPreferencesFromCsv(const std::string& file_path)
        {
            std::vector<std::vector<unsigned>> prefs;
            for (const auto& row : ReadCsvRows(file_path))
            {
                std::vector<unsigned> p;
                for (double v : row) p.push_back(static_cast<unsigned>(v));
                prefs.push_back(p);
            }
            return prefs;
        }
    } // namespace utils
} // namespace sequential_line_search
