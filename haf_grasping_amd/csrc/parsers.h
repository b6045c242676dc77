// parsers.h -- readers for the three construction-time files of the server (server.cpp:217-225) with the
// reference's parsing quirks preserved, and the PCD reader of the client path (client.cpp:141).
#pragma once

#include <string>
#include <vector>

namespace haf {

// Bounds applied to numbers read out of the files BEFORE they size an allocation: the engine itself takes at most 324
// attributes (kernels.h: kKP); these only keep a corrupt or hostile file from asking for gigabytes.
constexpr int kMaxAttributeIndex = 65536;
constexpr int kMaxSupportVectors = 1 << 22;

// data/Features.txt as CIntImage_to_Featurevec::read_features (fv.cpp:47-84) sees it
struct FeatureRow {
    int   reg[16];   // 4 regions x (x1, x2, y1, y2), inclusive cell coordinates in the 14x14 window
    float w[4];      // effective weights: w[3] is always 0 (4-region constructor never stores it, CHaarFeature.cpp:56-60)
};
bool load_features(const std::string &path, std::vector<FeatureRow> &rows, std::string &err);

// svm-scale range file, restore path (svm-scale.c:204-231)
struct RangeTable {
    double lower = -1.0, upper = 1.0;
    int max_index = 0;
    std::vector<double> fmin, fmax;        // [max_index + 1]
    std::vector<unsigned char> present;
};
bool load_range(const std::string &path, RangeTable &rt, std::string &err);

// libsvm text model, the subset the server exercises: 2-class C-SVC / nu-SVC with RBF kernel (svm.cpp:2714-2927)
enum { HAF_KERNEL_LINEAR = 0, HAF_KERNEL_POLY = 1, HAF_KERNEL_RBF = 2, HAF_KERNEL_SIGMOID = 3 };   // libsvm's kernel_type indices (svm.h)
struct SvmModel {
    int kernel_type = HAF_KERNEL_RBF;      // Kernel::k_function, svm.cpp:318-371 (precomputed kernels have no attribute vectors: refused)
    int degree = 0;                        // polynomial
    double coef0 = 0;                      // polynomial, sigmoid
    double gamma = 0, rho = 0;
    int n_sv = 0, dim = 0;                 // dim = largest attribute index
    int n_sv_class[2] = {0, 0};
    bool has_prob = false;          // probA and probB both present (svm.cpp:2811-2824; svm_check_probability_model 3098-3104)
    double probA = 0.0, probB = 0.0;
    int label[2] = {0, 0};
    std::vector<double> coef;              // [n_sv]
    std::vector<double> sv;                // dense [n_sv][dim], attribute k in column k-1
};
bool load_model(const std::string &path, SvmModel &m, std::string &err);

// PCD v0.7: ascii, binary, binary_compressed (LZF, SoA); x, y, z must be 4-byte floats
bool load_pcd(const std::string &path, std::vector<float> &xyz, std::string &err);

}  // namespace haf
