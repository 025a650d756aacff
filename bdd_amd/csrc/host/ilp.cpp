// ilp.cpp — see ilp.hpp.
#include "ilp.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <set>
#include <sstream>

namespace bddmma_host {

bool constraint::is_simplex() const
{
    if (ineq != ineq_t::eq || rhs == 0) return false;
    for (long c : coefficients)
        if (c != rhs) return false;
    return true;
}

size_t ilp_input::var(const std::string& name)
{
    auto it = index_.find(name);
    if (it != index_.end()) return it->second;
    const size_t i = var_names.size();
    index_.emplace(name, i);
    var_names.push_back(name);
    objective.push_back(0.0);
    return i;
}

double ilp_input::evaluate(const std::vector<char>& x) const
{
    double s = constant;
    for (size_t i = 0; i < objective.size(); ++i) s += objective[i] * (x[i] ? 1.0 : 0.0);
    return s;
}

bool ilp_input::feasible(const std::vector<char>& x) const
{
    for (const auto& c : constraints) {
        long s = 0;
        for (size_t i = 0; i < c.variables.size(); ++i) s += c.coefficients[i] * (x[c.variables[i]] ? 1 : 0);
        const bool ok = c.ineq == ineq_t::le ? s <= c.rhs : c.ineq == ineq_t::eq ? s == c.rhs : s >= c.rhs;
        if (!ok) return false;
    }
    return true;
}

ilp_input ilp_input::reduce(const std::set<size_t>& zeros, const std::set<size_t>& ones) const
{
    ilp_input r;
    r.constant = constant;
    std::vector<size_t> map(nr_variables(), SIZE_MAX);
    for (size_t i = 0; i < nr_variables(); ++i) {
        if (zeros.count(i) && ones.count(i)) throw std::runtime_error("variable '" + var_names[i] + "' is fixed to 0 and to 1");
        if (ones.count(i)) r.constant += objective[i];
        if (!zeros.count(i) && !ones.count(i)) {
            map[i] = r.var(var_names[i]);
            r.objective[map[i]] = objective[i];
        }
    }
    for (const auto& c : constraints) {
        constraint n;
        n.name = c.name;
        n.ineq = c.ineq;
        n.rhs = c.rhs;
        for (size_t k = 0; k < c.variables.size(); ++k) {
            const size_t v = c.variables[k];
            if (zeros.count(v)) continue;
            if (ones.count(v)) { n.rhs -= c.coefficients[k]; continue; }
            n.coefficients.push_back(c.coefficients[k]);
            n.variables.push_back(map[v]);
        }
        if (!n.variables.empty()) { r.constraints.push_back(std::move(n)); continue; }
        // nothing left: the row reads `0 {<=,=,>=} rhs`
        const bool ok = c.ineq == ineq_t::le ? 0 <= n.rhs : c.ineq == ineq_t::eq ? 0 == n.rhs : 0 >= n.rhs;
        if (!ok) throw std::runtime_error("reduced model not feasible due to violated constraint " + c.name);
    }
    return r;
}

void ilp_input::normalize()
{
    for (auto& c : constraints) {
        std::vector<size_t> idx(c.variables.size());
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return c.variables[a] < c.variables[b]; });
        constraint n = c;
        for (size_t i = 0; i < idx.size(); ++i) { n.variables[i] = c.variables[idx[i]]; n.coefficients[i] = c.coefficients[idx[i]]; }
        c = std::move(n);
    }
}

static std::string fmt_mag(double m)
{
    if (m == std::floor(m) && std::fabs(m) < 1e15) {
        std::ostringstream o;
        o << (long long)m;
        return o.str();
    }
    std::ostringstream o;
    o.precision(17);
    o << m;
    return o.str();
}

std::string ilp_input::write_lp() const
{
    auto term = [](double c, const std::string& name, bool first) {
        std::string s = c < 0 ? "- " : (first ? "" : "+ ");
        return s + fmt_mag(std::fabs(c)) + " " + name;
    };
    std::ostringstream o;
    o << "Minimize\n";
    for (size_t i = 0; i < objective.size(); ++i) o << term(objective[i], var_names[i], i == 0) << ((i % 8 == 7 || i + 1 == objective.size()) ? "\n" : " ");
    o << "Subject To\n";
    for (const auto& c : constraints) {
        if (!c.name.empty()) o << c.name << ": ";
        for (size_t i = 0; i < c.variables.size(); ++i) o << term((double)c.coefficients[i], var_names[c.variables[i]], i == 0) << " ";
        o << (c.ineq == ineq_t::le ? "<=" : c.ineq == ineq_t::eq ? "=" : ">=") << " " << c.rhs << "\n";
    }
    o << "End\n";
    return o.str();
}

// ------------------------------------------------------------------------------------------- scanner
namespace {

bool is_var_start(char c) { return std::isalpha((unsigned char)c); }
bool is_var_char(char c)
{
    return std::isalnum((unsigned char)c) || std::string("_-/(){},#;[].'").find(c) != std::string::npos;
}

std::string lower(std::string s)
{
    for (auto& c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}

std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

// number: digits[.digits*][e[+-]digits] | .digits[e..]; returns chars consumed (0 = no number at pos)
size_t scan_number(const std::string& s, size_t pos, double* out)
{
    size_t p = pos;
    size_t nd = 0;
    while (p < s.size() && std::isdigit((unsigned char)s[p])) { ++p; ++nd; }
    if (p < s.size() && s[p] == '.') {
        size_t q = p + 1, frac = 0;
        while (q < s.size() && std::isdigit((unsigned char)s[q])) { ++q; ++frac; }
        if (nd == 0 && frac == 0) return 0;
        p = q;
    } else if (nd == 0) {
        return 0;
    }
    if (p < s.size() && (s[p] == 'e' || s[p] == 'E')) {
        size_t q = p + 1;
        if (q < s.size() && (s[q] == '+' || s[q] == '-')) ++q;
        size_t ed = 0;
        while (q < s.size() && std::isdigit((unsigned char)s[q])) { ++q; ++ed; }
        if (ed > 0) p = q;
    }
    *out = std::strtod(s.substr(pos, p - pos).c_str(), nullptr);
    return p - pos;
}

struct term { double coeff; std::string name; };

// [sign] [number] [*] variable ...; a trailing "sign number" without a variable is the constant (objective only)
std::vector<term> scan_terms(const std::string& s, const char* what, bool allow_constant, double* constant)
{
    std::vector<term> out;
    size_t p = 0;
    auto skip = [&]() { while (p < s.size() && std::isspace((unsigned char)s[p])) ++p; };
    for (;;) {
        skip();
        if (p >= s.size()) break;
        const size_t start = p;
        double sign = 1.0;
        bool has_sign = false;
        if (s[p] == '+' || s[p] == '-') { sign = s[p] == '-' ? -1.0 : 1.0; has_sign = true; ++p; }
        if (!has_sign && !out.empty()) {
            // The reference grammar wants a sign in front of every term but the first (ILP_parser.cpp:54-60,100-104;
            // OPB_parser.cpp:43,57).  In a row `x y`, `x*y` and `2 x * y` are products of variables
            // (inequality_monomial, ILP_parser.cpp:86-98) which to_bdds() has no converter for: refused, not read as sums.
            if (std::string(what) == "constraint")
                throw std::runtime_error("nonlinear constraint (product of variables) near '" + s.substr(start, 40) + "' is not supported");
            throw std::runtime_error(std::string("cannot parse ") + what + ": term without a sign near '" + s.substr(start, 40) + "'");
        }
        skip();
        double num = 1.0;
        const size_t nn = scan_number(s, p, &num);
        p += nn;
        skip();
        if (p < s.size() && s[p] == '*') { ++p; skip(); }
        if (p < s.size() && is_var_start(s[p])) {
            const size_t v0 = p;
            while (p < s.size() && is_var_char(s[p])) ++p;
            out.push_back({sign * (nn ? num : 1.0), s.substr(v0, p - v0)});
            continue;
        }
        if (allow_constant && has_sign && nn && p >= s.size()) {
            *constant = sign * num;
            break;
        }
        throw std::runtime_error(std::string("cannot parse ") + what + " near '" + s.substr(start, 40) + "'");
    }
    return out;
}

// One line of the Bounds section.  The reference's four forms (ILP_parser.cpp:128-131: `x = v`, `x <= v`, `v <= x`,
// `lb <= x <= ub`, v in {0, 1}) and their mirror images with `>=`; its own test inputs hold `x2 >= 0`
// (test/test_ILP_parser.cpp:15).  Anything else is an error: the line would change the model if it meant something.
void scan_bound(const ilp_input& ilp, const std::string& line, std::set<size_t>& zeros, std::set<size_t>& ones)
{
    std::vector<std::string> tok;  // operands and relations in turn
    size_t p = 0;
    while (p < line.size()) {
        if (std::isspace((unsigned char)line[p])) { ++p; continue; }
        if ((line[p] == '<' || line[p] == '>') && p + 1 < line.size() && line[p + 1] == '=') { tok.push_back(line.substr(p, 2)); p += 2; continue; }
        if (line[p] == '=') { tok.push_back("="); ++p; continue; }
        size_t q = p;
        if (is_var_start(line[p])) while (q < line.size() && is_var_char(line[q])) ++q;
        else while (q < line.size() && !std::isspace((unsigned char)line[q]) && line[q] != '<' && line[q] != '>' && line[q] != '=') ++q;
        if (q == p) ++q;  // a lone `<` or `>`: its own token, refused below
        tok.push_back(line.substr(p, q - p));
        p = q;
    }
    const auto bad = [&](const std::string& why) { return std::runtime_error("cannot read Bounds line '" + line + "': " + why); };
    const auto is_rel = [](const std::string& t) { return t == "<=" || t == ">=" || t == "="; };
    const auto value = [&](const std::string& t) -> int {
        if (t == "0" || t == "+0" || t == "0.0") return 0;
        if (t == "1" || t == "+1" || t == "1.0") return 1;
        throw bad("a bound of a binary variable is 0 or 1");
    };
    const auto variable = [&](const std::string& t) -> size_t {
        if (!ilp.has_var(t)) throw bad("variable '" + t + "' is in no row and not in the objective");
        return ilp.var_index(t);
    };
    const auto lower = [&](size_t v, int b) { if (b == 1) ones.insert(v); };   // b <= x
    const auto upper = [&](size_t v, int b) { if (b == 0) zeros.insert(v); };  // x <= b
    size_t v = 0;
    if (tok.size() == 3 && is_rel(tok[1])) {
        const bool var_first = is_var_start(tok[0][0]);
        v = variable(var_first ? tok[0] : tok[2]);
        const int b = value(var_first ? tok[2] : tok[0]);
        const std::string& rel = tok[1];
        if (rel == "=") { lower(v, b); upper(v, b); }
        else if ((rel == "<=") == var_first) upper(v, b);
        else lower(v, b);
    } else if (tok.size() == 5 && tok[1] == tok[3] && (tok[1] == "<=" || tok[1] == ">=")) {
        v = variable(tok[2]);
        const int a = value(tok[0]), c = value(tok[4]);
        const int lb = tok[1] == "<=" ? a : c, ub = tok[1] == "<=" ? c : a;
        if (lb > ub) throw bad("lower bound above upper bound");
        lower(v, lb);
        upper(v, ub);
    } else {
        throw bad("expected `x = v`, `x <= v`, `x >= v`, `v <= x`, `v >= x` or `lb <= x <= ub`");
    }
    if (zeros.count(v) && ones.count(v)) throw bad("variable is fixed to 0 and to 1");
}

bool is_keyword_line(const std::string& line, std::initializer_list<const char*> words)
{
    const std::string t = lower(trim(line));
    for (const char* w : words)
        if (t == w) return true;
    return false;
}

}  // namespace

ilp_input parse_lp(const std::string& text)
{
    std::vector<std::string> lines;
    {
        std::istringstream in(text);
        std::string ln;
        while (std::getline(in, ln)) {
            if (!ln.empty() && ln.back() == '\r') ln.pop_back();
            const std::string t = trim(ln);
            if (!t.empty() && t[0] == '\\') continue;  // comment line
            lines.push_back(ln);
        }
    }
    size_t i_min = lines.size(), i_st = lines.size();
    for (size_t i = 0; i < lines.size(); ++i)
        if (is_keyword_line(lines[i], {"minimize", "minimise", "min"})) { i_min = i; break; }
    if (i_min == lines.size()) throw std::runtime_error("LP text has no 'Minimize' line");
    for (size_t i = i_min + 1; i < lines.size(); ++i)
        if (is_keyword_line(lines[i], {"subject to", "s.t.", "st"})) { i_st = i; break; }
    if (i_st == lines.size()) throw std::runtime_error("LP text has no 'Subject To' line");

    ilp_input ilp;
    std::string obj;
    for (size_t i = i_min + 1; i < i_st; ++i) obj += lines[i] + " ";
    obj = trim(obj);
    {  // optional objective name "name:"
        size_t p = 0;
        if (p < obj.size() && (std::isalpha((unsigned char)obj[p]) || obj[p] == '_')) {
            size_t q = p;
            while (q < obj.size() && (std::isalnum((unsigned char)obj[q]) || obj[q] == '_')) ++q;
            size_t r = q;
            while (r < obj.size() && std::isspace((unsigned char)obj[r])) ++r;
            if (r < obj.size() && obj[r] == ':') obj = obj.substr(r + 1);
        }
    }
    for (const auto& t : scan_terms(obj, "objective", true, &ilp.constant)) ilp.objective[ilp.var(t.name)] += t.coeff;

    std::string pending;
    size_t i_sec = lines.size();
    for (size_t i = i_st + 1; i < lines.size(); ++i) {
        const std::string s = trim(lines[i]);
        if (s.empty()) continue;
        if (is_keyword_line(s, {"end", "bounds", "binaries", "binary", "generals", "general", "coalesce"})) { i_sec = i; break; }
        pending = trim(pending + " " + s);
        // relation: first of <=, >=, =
        size_t rp = std::string::npos, rl = 0;
        for (size_t p = 0; p < pending.size(); ++p) {
            if ((pending[p] == '<' || pending[p] == '>') && p + 1 < pending.size() && pending[p + 1] == '=') { rp = p; rl = 2; break; }
            if (pending[p] == '=') { rp = p; rl = 1; break; }
        }
        if (rp == std::string::npos) continue;
        const std::string rhs_s = trim(pending.substr(rp + rl));
        double rhs = 0;
        {  // [+-] number, nothing else — otherwise the right-hand side is not complete yet
            size_t p = 0;
            double sign = 1;
            if (p < rhs_s.size() && (rhs_s[p] == '+' || rhs_s[p] == '-')) { sign = rhs_s[p] == '-' ? -1 : 1; ++p; }
            while (p < rhs_s.size() && std::isspace((unsigned char)rhs_s[p])) ++p;
            const size_t nn = scan_number(rhs_s, p, &rhs);
            if (nn == 0 || p + nn != rhs_s.size()) continue;
            rhs *= sign;
        }
        std::string lhs = pending.substr(0, rp);
        constraint c;
        {  // optional row name "name:"
            const std::string t = trim(lhs);
            size_t q = 0;
            while (q < t.size() && !std::isspace((unsigned char)t[q]) && t[q] != ':') ++q;
            size_t r = q;
            while (r < t.size() && std::isspace((unsigned char)t[r])) ++r;
            if (q > 0 && r < t.size() && t[r] == ':') { c.name = t.substr(0, q); lhs = t.substr(r + 1); }
        }
        c.ineq = rl == 1 ? ineq_t::eq : pending[rp] == '<' ? ineq_t::le : ineq_t::ge;
        if (rhs != std::floor(rhs)) throw std::runtime_error("only integer constraint coefficients are supported (as the reference, ILP_parser.cpp:262-300)");
        c.rhs = (long)rhs;
        for (const auto& t : scan_terms(lhs, "constraint", false, nullptr)) {
            if (t.coeff != std::floor(t.coeff)) throw std::runtime_error("only integer constraint coefficients are supported (as the reference, ILP_parser.cpp:262-300)");
            c.coefficients.push_back((long)t.coeff);
            c.variables.push_back(ilp.var(t.name));
        }
        ilp.constraints.push_back(std::move(c));
        pending.clear();
    }
    if (!pending.empty()) throw std::runtime_error("incomplete constraint: '" + pending.substr(0, 60) + "'");

    // Sections behind the rows.  `Bounds` lines fix variables (ILP_parser.cpp:128-131,343-436); the lists of the other
    // sections are skipped, every variable being binary anyway (:144, `until<end_line>`).
    std::set<size_t> zeros, ones;
    bool in_bounds = false;
    for (size_t i = i_sec; i < lines.size(); ++i) {
        const std::string s = trim(lines[i]);
        if (s.empty()) continue;
        if (is_keyword_line(s, {"end"})) break;
        if (is_keyword_line(s, {"bounds"})) { in_bounds = true; continue; }
        if (is_keyword_line(s, {"binaries", "binary", "generals", "general", "coalesce"})) { in_bounds = false; continue; }
        if (in_bounds) scan_bound(ilp, s, zeros, ones);
    }
    if (!zeros.empty() || !ones.empty()) return ilp.reduce(zeros, ones);
    return ilp;
}

ilp_input parse_opb(const std::string& text)
{
    // drop the leading comment lines, then treat the rest as one stream of `;`-terminated statements
    std::string body;
    {
        std::istringstream in(text);
        std::string ln;
        bool header = true;
        while (std::getline(in, ln)) {
            if (!ln.empty() && ln.back() == '\r') ln.pop_back();
            if (header && !trim(ln).empty() && trim(ln)[0] == '*') continue;
            if (!trim(ln).empty()) header = false;
            body += ln + " ";
        }
    }
    const std::string t = trim(body);
    if (t.compare(0, 4, "min:") != 0) throw std::runtime_error("could not read input: OPB text must start with 'min:'");
    size_t p = t.find(';');
    if (p == std::string::npos) throw std::runtime_error("could not read input: objective is not terminated by ';'");
    ilp_input ilp;
    double dummy = 0;
    for (const auto& tm : scan_terms(t.substr(4, p - 4), "objective", false, &dummy)) ilp.objective[ilp.var(tm.name)] += tm.coeff;
    for (;;) {
        const size_t q = t.find(';', p + 1);
        if (q == std::string::npos) break;  // `until<eof>`: the remainder is not a complete row
        const std::string row = trim(t.substr(p + 1, q - p - 1));
        p = q;
        if (row.empty()) continue;
        size_t rp = std::string::npos, rl = 0;
        for (size_t i = 0; i < row.size(); ++i) {
            if ((row[i] == '<' || row[i] == '>') && i + 1 < row.size() && row[i + 1] == '=') { rp = i; rl = 2; break; }
            if (row[i] == '=') { rp = i; rl = 1; break; }
        }
        if (rp == std::string::npos) throw std::runtime_error("cannot parse constraint near '" + row.substr(0, 40) + "'");
        constraint c;
        c.ineq = rl == 1 ? ineq_t::eq : row[rp] == '<' ? ineq_t::le : ineq_t::ge;
        {
            const std::string rhs_s = trim(row.substr(rp + rl));
            size_t i = 0;
            long sign = 1;
            if (i < rhs_s.size() && (rhs_s[i] == '+' || rhs_s[i] == '-')) { sign = rhs_s[i] == '-' ? -1 : 1; ++i; }
            if (i >= rhs_s.size()) throw std::runtime_error("cannot parse constraint near '" + row.substr(0, 40) + "'");
            long v = 0;
            for (; i < rhs_s.size(); ++i) {
                if (!std::isdigit((unsigned char)rhs_s[i])) throw std::runtime_error("only integer constraint coefficients are supported (OPB_parser.cpp:55)");
                v = v * 10 + (rhs_s[i] - '0');
            }
            c.rhs = sign * v;
        }
        for (const auto& tm : scan_terms(row.substr(0, rp), "constraint", false, nullptr)) {
            if (tm.coeff != std::floor(tm.coeff)) throw std::runtime_error("only integer constraint coefficients are supported (OPB_parser.cpp:47-48)");
            c.coefficients.push_back((long)tm.coeff);
            c.variables.push_back(ilp.var(tm.name));
        }
        ilp.constraints.push_back(std::move(c));
    }
    return ilp;
}

ilp_input parse_lp_or_opb(const std::string& text)
{
    try {
        return parse_lp(text);
    } catch (const std::exception& lp_error) {
        try {
            return parse_opb(text);
        } catch (const std::exception&) {
            throw std::runtime_error(lp_error.what());  // report the .lp diagnosis: it is the primary format
        }
    }
}

bdd_store to_bdds(const ilp_input& ilp)
{
    bdd_store col;
    for (const auto& c : ilp.constraints) {
        if (std::set<size_t>(c.variables.begin(), c.variables.end()).size() != c.variables.size())
            throw std::runtime_error("constraint '" + c.name + "' repeats a variable");
        if (c.is_simplex()) { col.add_simplex(c.variables); continue; }
        const row_status st = col.add_linear(c.coefficients, c.ineq, c.rhs, c.variables);
        if (st == row_status::infeasible) throw std::runtime_error("problem is infeasible");
    }
    return col;
}

}  // namespace bddmma_host
