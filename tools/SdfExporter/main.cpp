// SdfExporter — builds an OctreeSdf / ExactOctreeSdf from a mesh file on the MI355X and writes it in the reference's .bin layout.
// Same command line as the reference tool (src/tools/SdfExporter/main.cpp:28-48: positional model_path, output_path; -d/--depth,
// --start_depth, --termination_rule, --termination_threshold, --termination_threshold_by_distance, --min_triangles_per_node,
// --sdf_format octree|exact_octree, --algorithm uniform|no_continuity|continuity, -n/--normalize, --bb_margin, --num_threads) and
// the same defaults (octree: depth 8, start depth 1, CONTINUITY, trapezoidal rule 1e-3; exact octree: depth 5, start depth 1,
// 32 triangles per node; margin 20 %).  --sdf_format grid (UniformGridSdf) is outside the accelerated path.
//
// build:  g++ -std=c++17 -O2 -ffp-contract=off -I include tools/SdfExporter/main.cpp -Lsdflib_amd -lsdfhip -Wl,-rpath,$PWD/sdflib_amd -o SdfExporter
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "SdfLib/OctreeSdf.h"
#include "SdfLib/ExactOctreeSdf.h"

using namespace sdflib;

static void usage(const char* exe) {
    std::fprintf(stderr,
        "SdfExporter export an sdf\n  %s model_path output_path [-d depth] [--start_depth n] [--termination_rule trapezoidal_rule|simpsons_rule|by_distance_rule|none]\n"
        "      [--termination_threshold t] [--termination_threshold_by_distance t] [--min_triangles_per_node n] [--sdf_format octree|exact_octree]\n"
        "      [--algorithm uniform|no_continuity|continuity] [-n|--normalize] [--bb_margin percent] [--num_threads n]\n"
        "      [--devices 0,1,...|all]   (addition: build on several GPUs of this node; same as SDFLIB_DEVICES in the environment)\n", exe);
}

int main(int argc, char** argv) {
    std::vector<std::string> positional; std::map<std::string, std::string> opt; bool normalize = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
        if (a == "-n" || a == "--normalize") { normalize = true; continue; }
        if (a == "-d") a = "--depth";
        if (a == "-c") a = "--cell_size";
        if (a.rfind("--", 0) == 0) {
            std::string key = a.substr(2), value;
            const size_t eq = key.find('=');
            if (eq != std::string::npos) { value = key.substr(eq + 1); key = key.substr(0, eq); }
            else if (i + 1 < argc) value = argv[++i];
            else { std::fprintf(stderr, "Flag '%s' requires a value\n", key.c_str()); usage(argv[0]); return 1; }
            opt[key] = value;
        } else positional.push_back(a);
    }
    if (positional.empty()) { std::fprintf(stderr, "Error: No model_path specified\n"); usage(argv[0]); return 1; }
    if (opt.count("devices")) setenv("SDFLIB_DEVICES", opt["devices"].c_str(), 1);      // read by the classes' constructors (include/SdfLib/SdfFunction.h, detail::defaultMulti)
    auto has = [&](const char* k) { return opt.count(k) != 0; };
    auto num = [&](const char* k, double dflt) { return has(k) ? std::atof(opt[k].c_str()) : dflt; };
    const std::string sdfFormat = has("sdf_format") ? opt["sdf_format"] : "octree";
    const std::string modelPath = positional[0];
    const std::string outputPath = positional.size() > 1 ? positional[1] : "../output/sdfOctreeBunny.bin";

    Mesh mesh(modelPath);
    if (mesh.getVertices().empty()) return 1;
    BoundingBox box = mesh.getBoundingBox();
    if (normalize) {                                        // model units: largest extent -> 2, centred (reference :83-90)
        const glm::vec3 boxSize = box.getSize();
        const float maxSize = glm::max(glm::max(boxSize.x, boxSize.y), boxSize.z);
        mesh.applyTransform(glm::scale(glm::mat4(1.0f), glm::vec3(2.0f / maxSize)) * glm::translate(glm::mat4(1.0f), -box.getCenter()));
        box = mesh.getBoundingBox();
    }
    const glm::vec3 modelBBSize = box.getSize();
    const float margin = (float)num("bb_margin", 20.0) / 100.0f;
    box.addMargin(margin * glm::max(glm::max(modelBBSize.x, modelBBSize.y), modelBBSize.z));

    const auto t0 = std::chrono::steady_clock::now();
    std::unique_ptr<SdfFunction> sdfFunc;
    try {
        if (sdfFormat == "octree") {
            const std::string algorithm = has("algorithm") ? opt["algorithm"] : "continuity";
            OctreeSdf::InitAlgorithm initAlgorithm;
            if (algorithm == "uniform") initAlgorithm = OctreeSdf::InitAlgorithm::UNIFORM;
            else if (algorithm == "no_continuity") initAlgorithm = OctreeSdf::InitAlgorithm::NO_CONTINUITY;
            else if (algorithm == "continuity") initAlgorithm = OctreeSdf::InitAlgorithm::CONTINUITY;
            else { std::fprintf(stderr, "%s is not a valid supported octree generation algorithm\n", algorithm.c_str()); return 0; }
            const std::optional<OctreeSdf::TerminationRule> ruleOpt = OctreeSdf::stringToTerminationRule(has("termination_rule") ? opt["termination_rule"] : "trapezoidal_rule");
            if (!ruleOpt) { std::fprintf(stderr, "%s is not a valid termination rule\n", opt["termination_rule"].c_str()); return 0; }
            const OctreeSdf::TerminationRule rule = ruleOpt.value();
            OctreeSdf::TerminationRuleParams params = OctreeSdf::TerminationRuleParams::setNoneRuleParams();
            const float thr = (float)num("termination_threshold", 1e-3);
            if (rule == OctreeSdf::TerminationRule::TRAPEZOIDAL_RULE || rule == OctreeSdf::TerminationRule::SIMPSONS_RULE) params = OctreeSdf::TerminationRuleParams::setTrapezoidalRuleParams(thr);
            else if (rule == OctreeSdf::TerminationRule::BY_DISTANCE_RULE) params = OctreeSdf::TerminationRuleParams::setByDistanceRuleParams(thr, (float)num("termination_threshold_by_distance", 0.0));
            sdfFunc.reset(new OctreeSdf(mesh, box, (uint32_t)num("depth", 8), (uint32_t)num("start_depth", 1), rule, params, initAlgorithm, (uint32_t)num("num_threads", 1)));
        } else if (sdfFormat == "exact_octree") {
            sdfFunc.reset(new ExactOctreeSdf(mesh, box, (uint32_t)num("depth", 5), (uint32_t)num("start_depth", 1), (uint32_t)num("min_triangles_per_node", 32), (uint32_t)num("num_threads", 1)));
        } else { std::fprintf(stderr, "The sdf_format can only be octree or exact_octree\n"); return 1; }
    } catch (const std::exception& e) { std::fprintf(stderr, "[error] %s\n", e.what()); return 1; }
    std::fprintf(stderr, "[info] Computation time %gs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    std::fprintf(stderr, "[info] Saving the model\n");
    return sdfFunc->saveToFile(outputPath) ? 0 : 1;
}
