// SdfError — accuracy / throughput report of an approximate SDF against an exact one, on device batches.
// Same inputs and reported figures as the reference tool (src/tools/SdfError/main.cpp:20-95: two .bin paths and a sample
// count in millions; µs per query of each, RMSE, MAE, max error), but the per-sample loops are the batched getDistances()
// calls of include/SdfLib/SdfFunction.h, so the timing is the MI355X path.  Samples: uniform in the approximate SDF's
// sample area shrunk by 1e-5 (reference :47-56), drawn from std::mt19937 instead of rand().
//
// build:  g++ -std=c++17 -O2 -ffp-contract=off -I include tools/SdfError/main.cpp -Lsdflib_amd -lsdfhip -Wl,-rpath,$PWD/sdflib_amd -o SdfError
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "SdfLib/OctreeSdf.h"
#include "SdfLib/ExactOctreeSdf.h"

int main(int argc, char** argv) {
    if (argc < 3 || std::string(argv[1]) == "-h" || std::string(argv[1]) == "--help") {
        std::fprintf(stderr, "Calculate the error of a sdf\n  %s sdf_path exact_sdf_path [num_samples_in_millions]\n", argv[0]);
        return 0;
    }
    std::unique_ptr<sdflib::SdfFunction> sdf, exact;
    try {
        sdf = sdflib::SdfFunction::loadFromFile(argv[1]);
        exact = sdflib::SdfFunction::loadFromFile(argv[2]);
    } catch (const std::exception& e) { std::fprintf(stderr, "[error] %s\n", e.what()); return 1; }
    if (!sdf || !exact) return 1;
    std::fprintf(stderr, "[info] Models Loaded\n");

    const size_t numSamples = 1000000ull * (size_t)(argc > 3 ? std::max(1, std::atoi(argv[3])) : 1);
    std::vector<glm::vec3> samples(numSamples);
    const sdflib::BoundingBox area = sdf->getSampleArea();
    const glm::vec3 center = area.getCenter(), size = area.getSize() - glm::vec3(1e-5f);
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> uni(0.0f, 1.0f);
    for (glm::vec3& p : samples) {
        const float x = uni(rng), y = uni(rng), z = uni(rng);
        p = glm::vec3(center.x + (x - 0.5f) * size.x, center.y + (y - 0.5f) * size.y, center.z + (z - 0.5f) * size.z);
    }

    auto timed = [&](const sdflib::SdfFunction& f, std::vector<float>& out) {
        f.getDistances(samples.data(), std::min<size_t>(numSamples, 1024), out.data());          // warm-up (module load, BVH upload)
        const auto t0 = std::chrono::steady_clock::now();
        f.getDistances(samples.data(), numSamples, out.data());
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    std::vector<float> sdfDist(numSamples), exactDist(numSamples);
    try {
        const double ts = timed(*sdf, sdfDist);
        std::printf("[info] Sdf us per query: %g (%g s, host buffers in and out)\n", ts * 1.0e6 / (double)numSamples, ts);
        const double te = timed(*exact, exactDist);
        std::printf("[info] Exact Sdf us per query: %g (%g s, host buffers in and out)\n", te * 1.0e6 / (double)numSamples, te);
    } catch (const std::exception& e) { std::fprintf(stderr, "[error] %s\n", e.what()); return 1; }

    double rmse = 0.0, mae = 0.0; float maxError = 0.0f;
    for (size_t s = 0; s < numSamples; s++) {
        const float diff = sdfDist[s] - exactDist[s];
        rmse += (double)(diff * diff);
        mae += (double)std::fabs(diff);
        maxError = std::fmax(maxError, std::fabs(diff));
    }
    rmse = std::sqrt(rmse / (double)numSamples);
    mae = mae / (double)numSamples;
    std::printf("[info] RMSE: %g\n[info] MAE: %g\n[info] Max error: %g\n", rmse, mae, (double)maxError);
    return 0;
}
