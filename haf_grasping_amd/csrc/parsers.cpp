// parsers.cpp -- see parsers.h.  Host-only, no device code.
#include "parsers.h"

#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace haf {

// ---------------------------------------------------------------------------------------------------
// Features.txt
// ---------------------------------------------------------------------------------------------------
namespace {

// The reference walks a line with `end = line.find("\t", start)` stored in an int and
// `line.substr(start, end - start)`, then `start = end + 1` (fv.cpp:67-76).  When no tab is left, end becomes
// -1: the field is "the rest of the line" and the cursor falls back to 0.  A cursor object reproduces exactly that,
// which is what turns the trailing EMPTY line of data/Features.txt into a 324th all-zero feature.
class TabCursor {
public:
    explicit TabCursor(const std::string &s) : s_(s) {}
    std::string next()
    {
        std::string field;
        size_t pos = (start_ <= (int)s_.size()) ? s_.find('\t', (size_t)start_) : std::string::npos;
        int end = (pos == std::string::npos) ? -1 : (int)pos;
        if (start_ <= (int)s_.size()) {
            size_t count = (size_t)(end - start_);            // wraps to "everything" when end == -1
            field = s_.substr((size_t)start_, count);
        }
        start_ = end + 1;
        return field;
    }
private:
    const std::string &s_;
    int start_ = 0;
};

}  // namespace

bool load_features(const std::string &path, std::vector<FeatureRow> &rows, std::string &err)
{
    std::ifstream in(path.c_str());
    if (!in) { err = "cannot open feature file " + path; return false; }
    rows.clear();
    std::string line;
    // A line is consumed only while the stream is still good() AFTER reading it: a last line without '\n' sets
    // eofbit and is dropped, a last EMPTY line terminated by '\n' is parsed as sixteen zeros (fv.cpp:60-82).
    for (std::getline(in, line); in.good(); std::getline(in, line)) {
        FeatureRow r;
        TabCursor cur(line);
        for (int i = 0; i < 16; i++) r.reg[i] = atoi(cur.next().c_str());
        float w[4];
        for (int j = 0; j < 4; j++) w[j] = (float)atof(cur.next().c_str());
        r.w[0] = w[0]; r.w[1] = w[1]; r.w[2] = w[2];
        r.w[3] = 0.0f;   // CHaarFeature's 4-region constructor leaves weights[3] at its zero initialisation
        rows.push_back(r);
    }
    if (rows.empty()) { err = "no feature rows in " + path; return false; }
    return true;
}

// ---------------------------------------------------------------------------------------------------
// range file
// ---------------------------------------------------------------------------------------------------
bool load_range(const std::string &path, RangeTable &rt, std::string &err)
{
    FILE *fp = fopen(path.c_str(), "r");
    if (!fp) { err = "cannot open range file " + path; return false; }
    rt = RangeTable();
    // same stdio conversions as svm-scale so that every decimal becomes the same double (svm-scale.c:210-229)
    int c = fgetc(fp);
    if (c == 'y') {
        double a, b;
        if (fscanf(fp, "%lf %lf\n", &a, &b) != 2 || fscanf(fp, "%lf %lf\n", &a, &b) != 2) {
            fclose(fp); err = "malformed y section in " + path; return false;
        }
    } else if (c != EOF) {
        ungetc(c, fp);
    }
    if (fgetc(fp) != 'x') { fclose(fp); err = "range file " + path + " has no x section"; return false; }
    if (fscanf(fp, "%lf %lf\n", &rt.lower, &rt.upper) != 2) { fclose(fp); err = "range file: missing lower/upper"; return false; }
    int idx; double lo, hi;
    while (fscanf(fp, "%d %lf %lf\n", &idx, &lo, &hi) == 3) {
        if (idx < 0) continue;
        // the table is indexed by attribute: bound it BEFORE it sizes an allocation (a corrupt line must not ask for 16 GiB)
        if (idx > kMaxAttributeIndex) {
            fclose(fp);
            err = "range file " + path + ": attribute index " + std::to_string(idx) + " exceeds " + std::to_string(kMaxAttributeIndex);
            return false;
        }
        if (idx >= (int)rt.fmin.size()) {
            rt.fmin.resize((size_t)idx + 1, 0.0);
            rt.fmax.resize((size_t)idx + 1, 0.0);
            rt.present.resize((size_t)idx + 1, 0);
        }
        rt.fmin[idx] = lo; rt.fmax[idx] = hi; rt.present[idx] = 1;
        if (idx > rt.max_index) rt.max_index = idx;
    }
    fclose(fp);
    if (!(rt.upper > rt.lower)) { err = "range file: upper <= lower"; return false; }   // svm-scale.c:69-73
    if (rt.fmin.empty()) { rt.fmin.assign(1, 0.0); rt.fmax.assign(1, 0.0); rt.present.assign(1, 0); }
    return true;
}

// ---------------------------------------------------------------------------------------------------
// libsvm model
// ---------------------------------------------------------------------------------------------------
bool load_model(const std::string &path, SvmModel &m, std::string &err)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) { err = "cannot open model file " + path; return false; }
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string text = ss.str();
    m = SvmModel();
    size_t pos = 0;
    auto token = [&](std::string &out) -> bool {
        while (pos < text.size() && isspace((unsigned char)text[pos])) pos++;
        size_t b = pos;
        while (pos < text.size() && !isspace((unsigned char)text[pos])) pos++;
        out = text.substr(b, pos - b);
        return !out.empty();
    };
    auto number = [&](double &v) -> bool { std::string t; if (!token(t)) return false; char *e; v = strtod(t.c_str(), &e); return e != t.c_str(); };
    auto integer = [&](int &v) -> bool { std::string t; if (!token(t)) return false; char *e; v = (int)strtol(t.c_str(), &e, 10); return e != t.c_str(); };

    std::string key, val;
    int nr_class = -1;
    bool body = false, have_a = false, have_b = false;
    while (token(key)) {
        if (key == "svm_type") {
            if (!token(val) || (val != "c_svc" && val != "nu_svc")) { err = "model: svm_type '" + val + "' is not a classifier the server path uses"; return false; }
        } else if (key == "kernel_type") {
            // (round 5: svm-predict serves any of libsvm's four vector kernels; the reference's own model is RBF, which is the one the
            // fast tiers are built for -- the others go through the libsvm-order tier for every evaluation, engine_request.cpp)
            if (!token(val)) { err = "model: kernel_type without a value"; return false; }
            if (val == "linear") m.kernel_type = HAF_KERNEL_LINEAR;
            else if (val == "polynomial") m.kernel_type = HAF_KERNEL_POLY;
            else if (val == "rbf") m.kernel_type = HAF_KERNEL_RBF;
            else if (val == "sigmoid") m.kernel_type = HAF_KERNEL_SIGMOID;
            else { err = "model: kernel_type '" + val + "' has no attribute vectors to score (linear, polynomial, rbf, sigmoid are served)"; return false; }
        } else if (key == "gamma") { if (!number(m.gamma)) { err = "model: bad gamma"; return false; } }
        else if (key == "degree") { if (!integer(m.degree) || m.degree < 0 || m.degree > 64) { err = "model: bad degree"; return false; } }
        else if (key == "coef0") { if (!number(m.coef0)) { err = "model: bad coef0"; return false; } }
        else if (key == "nr_class") { if (!integer(nr_class) || nr_class != 2) { err = "model: nr_class must be 2"; return false; } }
        else if (key == "total_sv") { if (!integer(m.n_sv) || m.n_sv <= 0 || m.n_sv > kMaxSupportVectors) { err = "model: bad total_sv (must be in [1, " + std::to_string(kMaxSupportVectors) + "])"; return false; } }
        else if (key == "rho") { if (nr_class != 2 || !number(m.rho)) { err = "model: bad rho"; return false; } }
        else if (key == "label") { if (nr_class != 2 || !integer(m.label[0]) || !integer(m.label[1])) { err = "model: bad label"; return false; } }
        else if (key == "probA") { if (nr_class != 2 || !number(m.probA)) { err = "model: bad probA"; return false; } have_a = true; }
        else if (key == "probB") { if (nr_class != 2 || !number(m.probB)) { err = "model: bad probB"; return false; } have_b = true; }
        else if (key == "nr_sv") { if (nr_class != 2 || !integer(m.n_sv_class[0]) || !integer(m.n_sv_class[1])) { err = "model: bad nr_sv"; return false; } }
        else if (key == "SV") {
            while (pos < text.size() && text[pos] != '\n') pos++;     // rest of the SV line (svm.cpp:2834-2838)
            if (pos < text.size()) pos++;
            body = true;
            break;
        } else { err = "model: unknown text in model file: [" + key + "]"; return false; }   // svm.cpp:2841-2852
    }
    if (!body || nr_class != 2 || m.n_sv <= 0) { err = "model: incomplete header"; return false; }
    m.has_prob = have_a && have_b;
    if (m.n_sv_class[0] + m.n_sv_class[1] != m.n_sv) { err = "model: nr_sv does not add up to total_sv"; return false; }

    // body: one line per SV: coef idx:val idx:val ...   (svm.cpp:2890-2916)
    struct Entry { int sv, idx; double val; };
    std::vector<Entry> entries;
    if ((size_t)m.n_sv > (text.size() - std::min(pos, text.size())) / 2 + 1) { err = "model: fewer SV lines than total_sv"; return false; }
    m.coef.assign((size_t)m.n_sv, 0.0);
    int maxidx = 0;
    for (int i = 0; i < m.n_sv; i++) {
        if (pos >= text.size()) { err = "model: fewer SV lines than total_sv"; return false; }
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        const char *p = text.c_str() + pos, *end = text.c_str() + eol;
        char *q;
        m.coef[(size_t)i] = strtod(p, &q);
        if (q == p) { err = "model: bad coefficient"; return false; }
        p = q;
        while (p < end) {
            while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
            if (p >= end) break;
            long idx = strtol(p, &q, 10);
            if (q == p || *q != ':') { err = "model: bad idx:val pair"; return false; }
            p = q + 1;
            double v = strtod(p, &q);
            if (q == p) { err = "model: bad attribute value"; return false; }
            p = q;
            if (idx < 1) { err = "model: attribute index < 1 (precomputed kernels are not supported)"; return false; }
            if (idx > kMaxAttributeIndex) { err = "model: attribute index " + std::to_string(idx) + " exceeds " + std::to_string(kMaxAttributeIndex); return false; }
            entries.push_back({i, (int)idx, v});
            if (idx > maxidx) maxidx = (int)idx;
        }
        pos = eol + 1;
    }
    if (maxidx <= 0) { err = "model: support vectors carry no attributes"; return false; }
    m.dim = maxidx;
    if ((size_t)m.n_sv * (size_t)m.dim > ((size_t)1 << 28)) { err = "model: total_sv x attribute dimension exceeds 2^28 values"; return false; }
    m.sv.assign((size_t)m.n_sv * (size_t)m.dim, 0.0);
    for (const Entry &e : entries) m.sv[(size_t)e.sv * m.dim + (e.idx - 1)] = e.val;
    return true;
}

// ---------------------------------------------------------------------------------------------------
// PCD
// ---------------------------------------------------------------------------------------------------
namespace {

bool lzf_decompress(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len)
{
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        unsigned ctrl = in[ip++];
        if (ctrl < 32) {                       // literal run
            size_t n = ctrl + 1;
            if (ip + n > in_len || op + n > out_len) return false;
            memcpy(out + op, in + ip, n);
            ip += n; op += n;
        } else {                               // back reference
            size_t len = ctrl >> 5;
            if (len == 7) { if (ip >= in_len) return false; len += in[ip++]; }
            if (ip >= in_len) return false;
            size_t off = ((size_t)(ctrl & 0x1f) << 8) + in[ip++] + 1;
            len += 2;
            if (off > op || op + len > out_len) return false;
            for (size_t k = 0; k < len; k++, op++) out[op] = out[op - off];
        }
    }
    return op == out_len;
}

}  // namespace

bool load_pcd(const std::string &path, std::vector<float> &xyz, std::string &err)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) { err = "cannot open " + path; return false; }
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string raw = ss.str();
    std::vector<std::string> fields, types;
    std::vector<int> sizes, counts;
    long width = -1, height = -1, points = -1;
    std::string mode;
    size_t pos = 0;
    while (pos < raw.size()) {
        size_t nl = raw.find('\n', pos);
        if (nl == std::string::npos) nl = raw.size();
        std::string line = raw.substr(pos, nl - pos);
        pos = nl + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ls(line);
        std::string key, t;
        ls >> key;
        if (key == "FIELDS" || key == "COLUMNS") { while (ls >> t) fields.push_back(t); }
        else if (key == "SIZE") { while (ls >> t) sizes.push_back(atoi(t.c_str())); }
        else if (key == "TYPE") { while (ls >> t) types.push_back(t); }
        else if (key == "COUNT") { while (ls >> t) counts.push_back(atoi(t.c_str())); }
        else if (key == "WIDTH") ls >> width;
        else if (key == "HEIGHT") ls >> height;
        else if (key == "POINTS") ls >> points;
        else if (key == "DATA") { ls >> mode; break; }
    }
    if (fields.empty() || sizes.size() != fields.size() || types.size() != fields.size() || mode.empty()) {
        err = "malformed PCD header in " + path; return false;
    }
    if (counts.empty()) counts.assign(fields.size(), 1);
    if (counts.size() != fields.size()) { err = "malformed PCD header in " + path + " (COUNT)"; return false; }
    for (size_t i = 0; i < fields.size(); i++)
        if (sizes[i] <= 0 || sizes[i] > 8 || counts[i] <= 0 || counts[i] > (1 << 20)) { err = "PCD header: SIZE/COUNT out of range in " + path; return false; }
    if (points < 0) points = (width > 0 && height > 0) ? width * height : -1;   // POINTS wins over the row count, like PCL
    if (points < 0) { err = "PCD header without POINTS/WIDTH/HEIGHT"; return false; }
    int fx = -1, fy = -1, fz = -1;
    std::vector<size_t> offs(fields.size() + 1, 0);
    for (size_t i = 0; i < fields.size(); i++) {
        offs[i + 1] = offs[i] + (size_t)sizes[i] * (size_t)counts[i];
        if (fields[i] == "x") fx = (int)i;
        if (fields[i] == "y") fy = (int)i;
        if (fields[i] == "z") fz = (int)i;
    }
    if (fx < 0 || fy < 0 || fz < 0) { err = "PCD without x/y/z fields"; return false; }
    for (int f : {fx, fy, fz})
        if (sizes[(size_t)f] != 4 || types[(size_t)f] != "F" || counts[(size_t)f] != 1) { err = "x/y/z must be 4-byte floats"; return false; }
    const size_t rec = offs.back();
    // POINTS sizes the output: it must be backed by data.  ascii needs >= 2 bytes per row, binary `rec` bytes per point; a
    // compressed stream expands by at most 264/3 (LZF: a 3-byte back reference yields up to 264 bytes)
    const size_t remain = raw.size() - std::min(pos, raw.size());
    const size_t max_points = (mode == "ascii") ? remain / 2 + 1 : (mode == "binary") ? remain / rec : remain / rec * 100 + 1024;
    if ((unsigned long)points > max_points) { err = "PCD: POINTS " + std::to_string(points) + " exceeds what the file can hold"; return false; }
    xyz.assign((size_t)points * 3, 0.0f);
    if (mode == "ascii") {
        // column position of x/y/z among the whitespace separated tokens of a row
        std::vector<int> col(fields.size(), 0);
        for (size_t i = 1; i < fields.size(); i++) col[i] = col[i - 1] + counts[i - 1];
        long k = 0;
        while (k < points && pos < raw.size()) {
            size_t nl = raw.find('\n', pos);
            if (nl == std::string::npos) nl = raw.size();
            const char *p = raw.c_str() + pos, *end = raw.c_str() + nl;
            pos = nl + 1;
            int c = 0;
            bool any = false;
            while (p < end) {
                while (p < end && isspace((unsigned char)*p)) p++;
                if (p >= end) break;
                char *q;
                float v = strtof(p, &q);             // correctly rounded to float, as an istream >> float
                if (q == p) break;
                if (c == col[(size_t)fx]) xyz[(size_t)k * 3 + 0] = v;
                if (c == col[(size_t)fy]) xyz[(size_t)k * 3 + 1] = v;
                if (c == col[(size_t)fz]) xyz[(size_t)k * 3 + 2] = v;
                p = q; c++; any = true;
            }
            if (any) k++;
        }
        if (k != points) { err = "PCD ascii: fewer rows than POINTS"; return false; }
        return true;
    }
    if (mode == "binary") {
        if ((size_t)points * rec > remain) { err = "PCD binary: truncated"; return false; }   // (points <= remain / rec: no wrap)
        const char *base = raw.data() + pos;
        for (long k = 0; k < points; k++) {
            memcpy(&xyz[(size_t)k * 3 + 0], base + (size_t)k * rec + offs[(size_t)fx], 4);
            memcpy(&xyz[(size_t)k * 3 + 1], base + (size_t)k * rec + offs[(size_t)fy], 4);
            memcpy(&xyz[(size_t)k * 3 + 2], base + (size_t)k * rec + offs[(size_t)fz], 4);
        }
        return true;
    }
    if (mode == "binary_compressed") {
        if (pos + 8 > raw.size()) { err = "PCD compressed: truncated"; return false; }
        uint32_t csize, usize;
        memcpy(&csize, raw.data() + pos, 4);
        memcpy(&usize, raw.data() + pos + 4, 4);
        if ((size_t)csize > remain - 8 || (size_t)usize < (size_t)points * rec || (size_t)usize > (size_t)csize * 100 + 1024) { err = "PCD compressed: bad sizes"; return false; }
        std::vector<unsigned char> buf(usize);
        if (!lzf_decompress((const unsigned char *)raw.data() + pos + 8, csize, buf.data(), usize)) { err = "PCD compressed: LZF stream corrupt"; return false; }
        // structure of arrays: all values of field 0, then field 1, ...
        for (long k = 0; k < points; k++) {
            memcpy(&xyz[(size_t)k * 3 + 0], buf.data() + offs[(size_t)fx] * (size_t)points + (size_t)k * 4, 4);
            memcpy(&xyz[(size_t)k * 3 + 1], buf.data() + offs[(size_t)fy] * (size_t)points + (size_t)k * 4, 4);
            memcpy(&xyz[(size_t)k * 3 + 2], buf.data() + offs[(size_t)fz] * (size_t)points + (size_t)k * 4, 4);
        }
        return true;
    }
    err = "unsupported PCD DATA mode " + mode;
    return false;
}

}  // namespace haf
