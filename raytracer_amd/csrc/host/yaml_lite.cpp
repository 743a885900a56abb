#include "yaml_lite.h"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace rt {
namespace yaml {

namespace {

struct Line {
    int indent;
    std::string text;  // content without indentation, comments and trailing blanks
    int lineno;
};

[[noreturn]] void fail(int lineno, const std::string& msg) {
    throw ParseError("yaml: line " + std::to_string(lineno) + ": " + msg);
}

// Remove a trailing comment: '#' at start or preceded by whitespace, outside quotes.
std::string strip_comment(const std::string& s) {
    char q = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[i];
        if (q) {
            if (c == q) q = 0;
        } else if (c == '\'' || c == '"') {
            q = c;
        } else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) {
            return s.substr(0, i);
        }
    }
    return s;
}

std::string rtrim(std::string s) {
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r')) s.pop_back();
    return s;
}

std::vector<Line> split_lines(const std::string& text) {
    std::vector<Line> out;
    size_t pos = 0;
    int lineno = 0;
    while (pos <= text.size()) {
        size_t nl = text.find('\n', pos);
        if (nl == std::string::npos) nl = text.size();
        std::string raw = text.substr(pos, nl - pos);
        pos = nl + 1;
        ++lineno;
        int indent = 0;
        while ((size_t)indent < raw.size() && raw[indent] == ' ') ++indent;
        if ((size_t)indent < raw.size() && raw[indent] == '\t') fail(lineno, "tab used for indentation");
        std::string body = rtrim(strip_comment(raw.substr(indent)));
        if (body.empty()) continue;
        if (body == "---" && indent == 0) continue;
        out.push_back({indent, body, lineno});
    }
    return out;
}

// Net bracket depth of a flow fragment, ignoring quoted text.
int flow_depth(const std::string& s) {
    int d = 0;
    char q = 0;
    for (char c : s) {
        if (q) {
            if (c == q) q = 0;
        } else if (c == '\'' || c == '"') {
            q = c;
        } else if (c == '[' || c == '{') {
            ++d;
        } else if (c == ']' || c == '}') {
            --d;
        }
    }
    return d;
}

struct FlowParser {
    const std::string& s;
    size_t p = 0;
    int lineno;
    FlowParser(const std::string& str, int ln) : s(str), lineno(ln) {}

    void ws() {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n')) ++p;
    }
    NodePtr scalar(const char* stops) {
        ws();
        auto n = std::make_shared<Node>();
        n->kind = Node::Scalar;
        n->line = lineno;
        if (p < s.size() && (s[p] == '\'' || s[p] == '"')) {
            char q = s[p++];
            std::string v;
            while (true) {
                if (p >= s.size()) fail(lineno, "unterminated quoted scalar");
                if (s[p] == q) {
                    if (q == '\'' && p + 1 < s.size() && s[p + 1] == '\'') {  // '' escape
                        v.push_back('\'');
                        p += 2;
                        continue;
                    }
                    ++p;
                    break;
                }
                if (q == '"' && s[p] == '\\' && p + 1 < s.size()) {
                    char e = s[p + 1];
                    v.push_back(e == 'n' ? '\n' : e == 't' ? '\t' : e);
                    p += 2;
                    continue;
                }
                v.push_back(s[p++]);
            }
            n->scalar = v;
            n->quoted = true;
            return n;
        }
        size_t b = p;
        while (p < s.size() && !std::strchr(stops, s[p])) ++p;
        n->scalar = rtrim(s.substr(b, p - b));
        if (n->scalar.empty() || n->scalar == "~" || n->scalar == "null") n->kind = Node::Null;
        return n;
    }
    NodePtr value(const char* stops) {
        ws();
        if (p < s.size() && s[p] == '[') return seq();
        if (p < s.size() && s[p] == '{') return map();
        return scalar(stops);
    }
    NodePtr seq() {
        auto n = std::make_shared<Node>();
        n->kind = Node::Seq;
        n->line = lineno;
        ++p;  // [
        ws();
        if (p < s.size() && s[p] == ']') {
            ++p;
            return n;
        }
        while (true) {
            n->seq.push_back(value(",]"));
            ws();
            if (p >= s.size()) fail(lineno, "unterminated flow sequence");
            if (s[p] == ',') {
                ++p;
                ws();
                if (p < s.size() && s[p] == ']') {  // trailing comma
                    ++p;
                    return n;
                }
                continue;
            }
            if (s[p] == ']') {
                ++p;
                return n;
            }
            fail(lineno, "expected ',' or ']' in flow sequence");
        }
    }
    NodePtr map() {
        auto n = std::make_shared<Node>();
        n->kind = Node::Map;
        n->line = lineno;
        ++p;  // {
        ws();
        if (p < s.size() && s[p] == '}') {
            ++p;
            return n;
        }
        while (true) {
            NodePtr k = scalar(":,}");
            ws();
            if (p >= s.size() || s[p] != ':') fail(lineno, "expected ':' in flow mapping");
            ++p;
            NodePtr v = value(",}");
            n->map.emplace_back(k->scalar, v);
            ws();
            if (p >= s.size()) fail(lineno, "unterminated flow mapping");
            if (s[p] == ',') {
                ++p;
                ws();
                if (p < s.size() && s[p] == '}') {
                    ++p;
                    return n;
                }
                continue;
            }
            if (s[p] == '}') {
                ++p;
                return n;
            }
            fail(lineno, "expected ',' or '}' in flow mapping");
        }
    }
};

// Position of the ':' that ends a block-mapping key in `t` (followed by space or end), or npos.
size_t key_colon(const std::string& t) {
    if (t.empty() || t[0] == '[' || t[0] == '{') return std::string::npos;
    char q = 0;
    for (size_t i = 0; i < t.size(); ++i) {
        char c = t[i];
        if (q) {
            if (c == q) q = 0;
        } else if ((c == '\'' || c == '"') && i == 0) {
            q = c;
        } else if (c == ':' && (i + 1 == t.size() || t[i + 1] == ' ')) {
            return i;
        }
    }
    return std::string::npos;
}

struct BlockParser {
    std::vector<Line> lines;
    size_t pos = 0;

    // Parse an inline value that begins on the current line (already split off), joining following
    // lines while a flow collection is still open.
    NodePtr inline_value(std::string text, int lineno) {
        if (!text.empty() && (text[0] == '[' || text[0] == '{')) {
            int depth = flow_depth(text);
            while (depth > 0) {
                if (pos >= lines.size()) fail(lineno, "unterminated flow collection");
                text += "\n" + lines[pos].text;
                depth += flow_depth(lines[pos].text);
                ++pos;
            }
            FlowParser fp(text, lineno);
            NodePtr v = fp.value("");
            fp.ws();
            if (fp.p != text.size()) fail(lineno, "trailing characters after flow collection");
            return v;
        }
        if (text[0] == '|' || text[0] == '>' || text[0] == '&' || text[0] == '*' || text[0] == '!')
            fail(lineno, "unsupported YAML feature '" + text.substr(0, 1) + "'");
        FlowParser fp(text, lineno);
        NodePtr v = fp.scalar("");
        fp.ws();
        if (fp.p != text.size()) fail(lineno, "trailing characters after scalar");
        return v;
    }

    // Parse the node whose first line is lines[pos] and whose indentation is exactly `indent`.
    NodePtr block(int indent) {
        const Line& first = lines[pos];
        if (first.text == "-" || first.text.compare(0, 2, "- ") == 0) return sequence(indent);
        if (key_colon(first.text) != std::string::npos) return mapping(indent);
        std::string text = first.text;
        int ln = first.lineno;
        ++pos;
        return inline_value(text, ln);
    }

    NodePtr sequence(int indent) {
        auto n = std::make_shared<Node>();
        n->kind = Node::Seq;
        n->line = lines[pos].lineno;
        while (pos < lines.size() && lines[pos].indent == indent &&
               (lines[pos].text == "-" || lines[pos].text.compare(0, 2, "- ") == 0)) {
            Line& l = lines[pos];
            if (l.text == "-") {
                ++pos;
                if (pos < lines.size() && lines[pos].indent > indent)
                    n->seq.push_back(block(lines[pos].indent));
                else
                    n->seq.push_back(std::make_shared<Node>());
                continue;
            }
            // Re-interpret "- rest" as a line holding `rest` indented past the dash.
            size_t skip = 2;
            while (skip < l.text.size() && l.text[skip] == ' ') ++skip;
            l.indent = indent + (int)skip;
            l.text = l.text.substr(skip);
            n->seq.push_back(block(l.indent));
        }
        if (pos < lines.size() && lines[pos].indent > indent) fail(lines[pos].lineno, "bad indentation in sequence");
        return n;
    }

    NodePtr mapping(int indent) {
        auto n = std::make_shared<Node>();
        n->kind = Node::Map;
        n->line = lines[pos].lineno;
        while (pos < lines.size() && lines[pos].indent == indent) {
            const Line l = lines[pos];
            size_t c = key_colon(l.text);
            if (c == std::string::npos) fail(l.lineno, "expected 'key: value'");
            std::string key = rtrim(l.text.substr(0, c));
            if (key.size() >= 2 && (key[0] == '\'' || key[0] == '"') && key.back() == key[0])
                key = key.substr(1, key.size() - 2);
            std::string rest = l.text.substr(c + 1);
            size_t b = rest.find_first_not_of(" \t");
            rest = b == std::string::npos ? "" : rest.substr(b);
            ++pos;
            NodePtr v;
            if (!rest.empty()) {
                v = inline_value(rest, l.lineno);
            } else if (pos < lines.size() && lines[pos].indent > indent) {
                v = block(lines[pos].indent);
            } else if (pos < lines.size() && lines[pos].indent == indent &&
                       (lines[pos].text == "-" || lines[pos].text.compare(0, 2, "- ") == 0)) {
                v = sequence(indent);  // "key:\n- a\n- b" with the dashes at the key's indentation
            } else {
                v = std::make_shared<Node>();
            }
            for (auto& kv : n->map)
                if (kv.first == key) fail(l.lineno, "duplicate key '" + key + "'");
            n->map.emplace_back(key, v);
        }
        if (pos < lines.size() && lines[pos].indent > indent) fail(lines[pos].lineno, "bad indentation in mapping");
        return n;
    }
};

}  // namespace

bool Node::is_number() const {
    if (kind != Scalar || quoted || scalar.empty()) return false;
    const std::string& s = scalar;
    if (s == ".inf" || s == "-.inf" || s == "+.inf" || s == ".nan" || s == ".Inf" || s == ".NaN") return true;
    size_t i = 0;
    if (s[i] == '+' || s[i] == '-') ++i;
    size_t digits = 0, frac = 0;
    while (i < s.size() && std::isdigit((unsigned char)s[i])) ++i, ++digits;
    if (i < s.size() && s[i] == '.') {
        ++i;
        while (i < s.size() && std::isdigit((unsigned char)s[i])) ++i, ++frac;
    }
    if (digits + frac == 0) return false;
    if (i < s.size() && (s[i] == 'e' || s[i] == 'E')) {
        ++i;
        if (i < s.size() && (s[i] == '+' || s[i] == '-')) ++i;
        size_t ed = 0;
        while (i < s.size() && std::isdigit((unsigned char)s[i])) ++i, ++ed;
        if (ed == 0) return false;
    }
    return i == s.size();
}

double Node::as_double() const {
    if (!is_number()) throw ParseError("yaml: line " + std::to_string(line) + ": expected a number, got '" + scalar + "'");
    if (scalar == ".inf" || scalar == "+.inf" || scalar == ".Inf") return HUGE_VAL;
    if (scalar == "-.inf") return -HUGE_VAL;
    if (scalar == ".nan" || scalar == ".NaN") return std::strtod("nan", nullptr);
    return std::strtod(scalar.c_str(), nullptr);
}

NodePtr parse(const std::string& text) {
    BlockParser bp;
    bp.lines = split_lines(text);
    if (bp.lines.empty()) return std::make_shared<Node>();
    NodePtr root = bp.block(bp.lines[0].indent);
    if (bp.pos != bp.lines.size()) fail(bp.lines[bp.pos].lineno, "unexpected content");
    return root;
}

}  // namespace yaml
}  // namespace rt
