// A YAML-subset reader sized for the scene files: block mappings and sequences,
// flow sequences/mappings (possibly nested, possibly spanning lines), '#' comments, plain / single- /
// double-quoted scalars.  It replaces serde_yaml in load.rs:48.  No anchors, tags, block scalars or
// multi-document streams.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rt {
namespace yaml {

struct Node;
using NodePtr = std::shared_ptr<Node>;

struct Node {
    enum Kind { Null, Scalar, Seq, Map } kind = Null;
    std::string scalar;   // Scalar
    bool quoted = false;  // Scalar: was it quoted in the source (then it is always a string)
    std::vector<NodePtr> seq;
    std::vector<std::pair<std::string, NodePtr>> map;  // insertion order kept
    int line = 0;

    const Node* find(const std::string& key) const {
        for (auto& kv : map)
            if (kv.first == key) return kv.second.get();
        return nullptr;
    }
    // serde_yaml would type an unquoted scalar like 15, .12, -.45, 1e3 as a number.
    bool is_number() const;
    double as_double() const;  // throws if not a number
};

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

NodePtr parse(const std::string& text);

}  // namespace yaml
}  // namespace rt
