// dataset.hpp -- the dataset description the two CLIs accept: a single FASTA/FASTQ(.gz) file or a
// YAML list of libraries with the keys the reference's SequencingLibraryBase::yamlize maps
// (common/pipeline/library.cpp:78-87: type, orientation, "left reads", "right reads",
// "interlaced reads", "merged reads", "single reads"); relative paths are resolved against the
// YAML file's directory (library.cpp:135-155).  Only what these two tools need is parsed: the flat
// "- key: [a, b]" / "- key:\n    - a" forms that spades.py and the reference's configs write.
#pragma once

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace bbkhost {

inline std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}

inline std::string unquote(std::string s) {
    s = trim(s);
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
        s = s.substr(1, s.size() - 2);
    return s;
}

inline std::string dirname_of(const std::string &p) {
    size_t s = p.find_last_of('/');
    return s == std::string::npos ? std::string(".") : p.substr(0, s);
}

inline bool ends_with(const std::string &s, const std::string &suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// one library: its read files by kind, plus the two scalar keys that are passed through to an output dataset
struct DatasetLib {
    std::vector<std::string> v[5];  // left, right, interlaced, merged, single
    std::string type, orientation;
};
enum { LIB_LEFT = 0, LIB_RIGHT = 1, LIB_INTERLACED = 2, LIB_MERGED = 3, LIB_SINGLE = 4 };

inline bool load_dataset_libs(const std::string &path, std::vector<DatasetLib> &libs, std::string &err) {
    std::ifstream in(path);
    if (!in) {
        err = "cannot open dataset file " + path;
        return false;
    }
    const std::string dir = dirname_of(path);
    typedef DatasetLib Lib;
    static const char *keys[5] = {"left reads", "right reads", "interlaced reads", "merged reads", "single reads"};
    int cur_key = -1;
    std::string line;
    auto add = [&](int key, std::string f) {
        f = unquote(f);
        if (f.empty()) return;
        if (f[0] != '/') f = dir + "/" + f;
        libs.back().v[key].push_back(f);
    };
    while (std::getline(in, line)) {
        std::string t = trim(line);
        if (t.empty() || t[0] == '#') continue;
        bool new_item = false;
        if (t[0] == '-' && (t.size() == 1 || t[1] == ' ')) {
            // either a new library ("- key: ...") or a list element ("- path")
            std::string rest = trim(t.substr(1));
            if (rest.find(':') != std::string::npos && rest[0] != '/' && rest[0] != '.') {
                size_t indent = line.find('-');
                if (indent <= 1 || libs.empty()) { libs.emplace_back(); new_item = true; }
                t = rest;
            } else {
                if (cur_key >= 0 && !libs.empty()) add(cur_key, rest);
                continue;
            }
        }
        (void)new_item;
        size_t colon = t.find(':');
        if (colon == std::string::npos) continue;
        if (libs.empty()) libs.emplace_back();
        std::string key = unquote(t.substr(0, colon));
        std::string val = trim(t.substr(colon + 1));
        cur_key = -1;
        for (int i = 0; i < 5; ++i)
            if (key == keys[i]) cur_key = i;
        if (cur_key < 0) {  // type / orientation are kept for the output dataset; the rest is not needed on this path
            if (key == "type") libs.back().type = unquote(val);
            if (key == "orientation") libs.back().orientation = unquote(val);
            continue;
        }
        if (!val.empty() && val[0] == '[') {
            size_t e = val.find(']');
            std::string inner = val.substr(1, e == std::string::npos ? std::string::npos : e - 1);
            std::stringstream ss(inner);
            std::string item;
            while (std::getline(ss, item, ',')) add(cur_key, item);
            cur_key = -1;
        } else if (!val.empty()) {
            add(cur_key, val);
            cur_key = -1;
        }
    }
    size_t n = 0;
    for (const Lib &l : libs)
        for (int i = 0; i < 5; ++i) n += l.v[i].size();
    if (n == 0) {
        err = "no read files found in " + path;
        return false;
    }
    return true;
}

// Returns every read file of every library, in library order (left, right, interlaced, merged, single).
inline bool load_dataset_yaml(const std::string &path, std::vector<std::string> &files, std::string &err) {
    std::vector<DatasetLib> libs;
    if (!load_dataset_libs(path, libs, err)) return false;
    for (const DatasetLib &l : libs)
        for (int i = 0; i < 5; ++i)
            for (const std::string &f : l.v[i]) files.push_back(f);
    return true;
}

// Writes a dataset description with the keys of SequencingLibraryBase::yamlize (common/pipeline/library.cpp:78-87).
inline bool save_dataset_yaml(const std::string &path, const std::vector<DatasetLib> &libs) {
    std::ofstream out(path);
    if (!out) return false;
    static const char *keys[5] = {"left reads", "right reads", "interlaced reads", "merged reads", "single reads"};
    for (const DatasetLib &l : libs) {
        bool first = true;
        auto lead = [&]() -> const char * {
            const char *p = first ? "- " : "  ";
            first = false;
            return p;
        };
        if (!l.orientation.empty()) out << lead() << "orientation: \"" << l.orientation << "\"\n";
        if (!l.type.empty()) out << lead() << "type: \"" << l.type << "\"\n";
        for (int i = 0; i < 5; ++i) {
            if (l.v[i].empty()) continue;
            out << lead() << keys[i] << ":\n";
            for (const std::string &f : l.v[i]) out << "    - \"" << f << "\"\n";
        }
    }
    return (bool)out;
}

}  // namespace bbkhost
