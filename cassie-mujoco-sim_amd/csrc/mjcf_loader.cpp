/*
 * mjcf_loader.cpp -- MJCF-subset parser + model compiler (host C++).
 *
 * Covers exactly the elements/attributes the three in-scope models use
 * (reference model/cassie.xml, model/cassie_hfield.xml, model/cassie_tray_box.xml;
 * census in SURVEY.md App. A.1) and produces a cm::HostModel.  It stands in for
 * the reference's mj_loadXML call (reference src/cassiemujoco.c:851, :997).
 *
 * Unknown elements are ignored when they cannot influence the dynamics
 * (visual, asset textures/materials/meshes, cameras' optics, lights) and are a
 * hard error otherwise (e.g. tendons, welds, non-motor actuators), so a model
 * outside the subset fails loudly instead of simulating something else.
 */
#include "host_model.h"
#include "topo_static.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

namespace cm {
namespace {

/* ------------------------------------------------------------------ XML --- */
struct XmlNode {
    std::string tag;
    std::vector<std::pair<std::string, std::string>> attr;
    std::vector<std::unique_ptr<XmlNode>> kids;
    const std::string *get(const char *k) const {
        for (auto &a : attr)
            if (a.first == k) return &a.second;
        return nullptr;
    }
};

struct XmlParser {
    const std::string &s;
    size_t p = 0;
    std::string err;
    explicit XmlParser(const std::string &src) : s(src) {}
    void skip_ws() {
        while (p < s.size() && isspace((unsigned char)s[p])) ++p;
    }
    bool starts(const char *lit) const { return s.compare(p, strlen(lit), lit) == 0; }
    bool skip_misc() { /* whitespace, comments, <? ?>, <!DOCTYPE> and text */
        for (;;) {
            while (p < s.size() && s[p] != '<') ++p;
            if (p >= s.size()) return true;
            if (starts("<!--")) {
                size_t e = s.find("-->", p + 4);
                if (e == std::string::npos) { err = "unterminated comment"; return false; }
                p = e + 3;
            } else if (starts("<?")) {
                size_t e = s.find("?>", p + 2);
                if (e == std::string::npos) { err = "unterminated <?"; return false; }
                p = e + 2;
            } else if (starts("<!")) {
                size_t e = s.find('>', p);
                if (e == std::string::npos) { err = "unterminated <!"; return false; }
                p = e + 1;
            } else
                return true;
        }
    }
    std::string name() {
        size_t b = p;
        while (p < s.size() && (isalnum((unsigned char)s[p]) || s[p] == '_' || s[p] == '-' || s[p] == ':' ||
                                s[p] == '.'))
            ++p;
        return s.substr(b, p - b);
    }
    std::unique_ptr<XmlNode> element() {
        /* at '<' of an opening tag */
        ++p;
        std::unique_ptr<XmlNode> n(new XmlNode);
        n->tag = name();
        if (n->tag.empty()) { err = "bad tag name"; return nullptr; }
        for (;;) {
            skip_ws();
            if (p >= s.size()) { err = "eof in tag"; return nullptr; }
            if (s[p] == '/') {
                if (p + 1 < s.size() && s[p + 1] == '>') { p += 2; return n; }
                err = "bad '/'"; return nullptr;
            }
            if (s[p] == '>') { ++p; break; }
            std::string k = name();
            if (k.empty()) { err = "bad attribute in <" + n->tag + ">"; return nullptr; }
            skip_ws();
            if (p >= s.size() || s[p] != '=') { err = "missing '=' after " + k; return nullptr; }
            ++p;
            skip_ws();
            if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) { err = "missing quote for " + k; return nullptr; }
            char q = s[p++];
            size_t e = s.find(q, p);
            if (e == std::string::npos) { err = "unterminated attribute " + k; return nullptr; }
            n->attr.emplace_back(k, s.substr(p, e - p));
            p = e + 1;
        }
        /* children until </tag> */
        for (;;) {
            if (!skip_misc()) return nullptr;
            if (p >= s.size()) { err = "eof inside <" + n->tag + ">"; return nullptr; }
            if (starts("</")) {
                p += 2;
                std::string c = name();
                skip_ws();
                if (c != n->tag || p >= s.size() || s[p] != '>') { err = "mismatched </" + c + ">"; return nullptr; }
                ++p;
                return n;
            }
            auto k = element();
            if (!k) return nullptr;
            n->kids.push_back(std::move(k));
        }
    }
    std::unique_ptr<XmlNode> parse() {
        if (!skip_misc()) return nullptr;
        if (p >= s.size()) { err = "no root element"; return nullptr; }
        return element();
    }
};

/* ----------------------------------------------------------------- math --- */
typedef std::map<std::string, std::string> AttrMap;

bool parse_doubles(const std::string &v, double *out, int nmin, int nmax, int *ngot = nullptr) {
    std::istringstream is(v);
    int n = 0;
    double x;
    while (is >> x) {
        if (n < nmax) out[n] = x;
        ++n;
    }
    if (ngot) *ngot = n;
    return n >= nmin && n <= nmax;
}

void cross3(double *r, const double *a, const double *b) {
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}
double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
double norm3(const double *a) { return std::sqrt(dot3(a, a)); }
double normalize3(double *a) {
    double n = norm3(a);
    if (n < CM_MINVAL) { a[0] = 1; a[1] = a[2] = 0; }
    else { a[0] /= n; a[1] /= n; a[2] /= n; }
    return n;
}
void normalize4(double *q) {
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < CM_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
    else for (int i = 0; i < 4; ++i) q[i] /= n;
}
void mulquat(double *r, const double *a, const double *b) {
    double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                   a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                   a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                   a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
    memcpy(r, t, sizeof t);
}
void quat2mat(double *m, const double *q) {
    double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
    double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
    double q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
    m[0] = q00 + q11 - q22 - q33; m[1] = 2 * (q12 - q03);       m[2] = 2 * (q13 + q02);
    m[3] = 2 * (q12 + q03);       m[4] = q00 - q11 + q22 - q33; m[5] = 2 * (q23 - q01);
    m[6] = 2 * (q13 - q02);       m[7] = 2 * (q23 + q01);       m[8] = q00 - q11 - q22 + q33;
}
void rotvec(double *r, const double *m, const double *v) { /* r = M v */
    double t[3] = {m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
                   m[6] * v[0] + m[7] * v[1] + m[8] * v[2]};
    memcpy(r, t, sizeof t);
}
void rotvecT(double *r, const double *m, const double *v) { /* r = M^T v */
    double t[3] = {m[0] * v[0] + m[3] * v[1] + m[6] * v[2], m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
                   m[2] * v[0] + m[5] * v[1] + m[8] * v[2]};
    memcpy(r, t, sizeof t);
}
void rotvecquat(double *r, const double *v, const double *q) {
    double m[9];
    quat2mat(m, q);
    rotvec(r, m, v);
}
/* rotation matrix (row-major, columns = frame axes) -> unit quaternion */
void mat2quat(double *q, const double *m) {
    double tr = m[0] + m[4] + m[8];
    if (tr > 0) {
        double s = std::sqrt(tr + 1.0) * 2;
        q[0] = 0.25 * s; q[1] = (m[7] - m[5]) / s; q[2] = (m[2] - m[6]) / s; q[3] = (m[3] - m[1]) / s;
    } else if (m[0] > m[4] && m[0] > m[8]) {
        double s = std::sqrt(1.0 + m[0] - m[4] - m[8]) * 2;
        q[0] = (m[7] - m[5]) / s; q[1] = 0.25 * s; q[2] = (m[1] + m[3]) / s; q[3] = (m[2] + m[6]) / s;
    } else if (m[4] > m[8]) {
        double s = std::sqrt(1.0 + m[4] - m[0] - m[8]) * 2;
        q[0] = (m[2] - m[6]) / s; q[1] = (m[1] + m[3]) / s; q[2] = 0.25 * s; q[3] = (m[5] + m[7]) / s;
    } else {
        double s = std::sqrt(1.0 + m[8] - m[0] - m[4]) * 2;
        q[0] = (m[3] - m[1]) / s; q[1] = (m[2] + m[6]) / s; q[2] = (m[5] + m[7]) / s; q[3] = 0.25 * s;
    }
    normalize4(q);
}
/* quaternion taking the z axis onto vec */
void z2quat(double *q, const double *vec_in) {
    double v[3] = {vec_in[0], vec_in[1], vec_in[2]};
    normalize3(v);
    double z[3] = {0, 0, 1}, ax[3];
    cross3(ax, z, v);
    double s = norm3(ax);
    if (s < 1e-10) { ax[0] = 1; ax[1] = ax[2] = 0; }
    else { ax[0] /= s; ax[1] /= s; ax[2] /= s; }
    double ang = std::atan2(s, v[2]);
    q[0] = std::cos(ang / 2);
    double sn = std::sin(ang / 2);
    q[1] = ax[0] * sn; q[2] = ax[1] * sn; q[3] = ax[2] * sn;
}
/* eigen-decomposition of a symmetric 3x3 (Jacobi sweeps); eigenvalues sorted
 * decreasing, eigenvectors = columns of a right-handed rotation returned as quat */
void eig3(const double *A, double *eval, double *quat) {
    double a[3][3] = {{A[0], A[1], A[2]}, {A[3], A[4], A[5]}, {A[6], A[7], A[8]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 3; ++k) { /* A <- A G */
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) { /* A <- G^T A */
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int idx[3] = {0, 1, 2};
    double ev[3] = {a[0][0], a[1][1], a[2][2]};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (ev[idx[j]] > ev[idx[i]]) std::swap(idx[i], idx[j]);
    double R[9];
    for (int c = 0; c < 3; ++c) {
        eval[c] = ev[idx[c]];
        for (int r = 0; r < 3; ++r) R[3 * r + c] = v[r][idx[c]];
    }
    /* make right-handed */
    double c0[3] = {R[0], R[3], R[6]}, c1[3] = {R[1], R[4], R[7]}, c2[3];
    cross3(c2, c0, c1);
    R[2] = c2[0]; R[5] = c2[1]; R[8] = c2[2];
    mat2quat(quat, R);
}

/* ------------------------------------------------------------- defaults --- */
struct Defaults {
    std::map<std::string, AttrMap> el; /* element tag -> attribute defaults */
};

struct Compiler {
    HostModel &m;
    std::string &err;
    bool degree = true;
    bool inertiafromgeom_auto = true;
    std::map<std::string, Defaults> classes;
    std::vector<std::pair<double, double>> geom_massdens; /* per geom: explicit mass (<0 none), density */

    Compiler(HostModel &hm, std::string &e) : m(hm), err(e) {}

    bool fail(const std::string &msg) {
        err = msg;
        return false;
    }

    void read_defaults(const XmlNode &d, const std::string &cls, const std::string &parent) {
        Defaults def;
        if (!parent.empty()) def = classes[parent];
        for (auto &k : d.kids) {
            if (k->tag == "default") continue;
            AttrMap &am = def.el[k->tag];
            for (auto &a : k->attr) am[a.first] = a.second;
        }
        classes[cls] = def;
        for (auto &k : d.kids)
            if (k->tag == "default") {
                const std::string *c = k->get("class");
                read_defaults(*k, c ? *c : std::string("main"), cls);
            }
    }

    /* effective attributes of element n given the class context */
    AttrMap resolve(const XmlNode &n, const std::string &childclass, const char *deftag = nullptr) {
        std::string cls = "main";
        if (!childclass.empty()) cls = childclass;
        if (const std::string *c = n.get("class")) cls = *c;
        AttrMap out;
        auto ci = classes.find(cls);
        if (ci != classes.end()) {
            auto ei = ci->second.el.find(deftag ? deftag : n.tag.c_str());
            if (ei != ci->second.el.end()) out = ei->second;
        }
        for (auto &a : n.attr) out[a.first] = a.second;
        return out;
    }

    static const std::string *aget(const AttrMap &a, const char *k) {
        auto i = a.find(k);
        return i == a.end() ? nullptr : &i->second;
    }
    static bool abool(const AttrMap &a, const char *k, bool dflt) {
        const std::string *v = aget(a, k);
        if (!v) return dflt;
        return *v == "true";
    }
    bool avec(const AttrMap &a, const char *k, double *out, int n, const char *ctx) {
        const std::string *v = aget(a, k);
        if (!v) return true;
        if (!parse_doubles(*v, out, n, n)) return fail(std::string("bad '") + k + "' in " + ctx);
        return true;
    }

    /* orientation from xyaxes / quat / zaxis (only these appear in scope) */
    bool orientation(const AttrMap &a, double *quat, const char *ctx) {
        quat[0] = 1; quat[1] = quat[2] = quat[3] = 0;
        if (const std::string *v = aget(a, "quat")) {
            if (!parse_doubles(*v, quat, 4, 4)) return fail(std::string("bad quat in ") + ctx);
            normalize4(quat);
        } else if (const std::string *v = aget(a, "xyaxes")) {
            double xy[6];
            if (!parse_doubles(*v, xy, 6, 6)) return fail(std::string("bad xyaxes in ") + ctx);
            double x[3] = {xy[0], xy[1], xy[2]}, y[3] = {xy[3], xy[4], xy[5]}, z[3];
            normalize3(x);
            double d = dot3(x, y);
            for (int i = 0; i < 3; ++i) y[i] -= d * x[i];
            normalize3(y);
            cross3(z, x, y);
            double R[9] = {x[0], y[0], z[0], x[1], y[1], z[1], x[2], y[2], z[2]};
            mat2quat(quat, R);
        } else if (const std::string *v = aget(a, "zaxis")) {
            double z[3];
            if (!parse_doubles(*v, z, 3, 3)) return fail(std::string("bad zaxis in ") + ctx);
            z2quat(quat, z);
        } else if (aget(a, "euler") || aget(a, "axisangle")) {
            return fail(std::string("euler/axisangle orientation not in the supported MJCF subset: ") + ctx);
        }
        return true;
    }

    /* ---------------------------------------------------------- bodies --- */
    bool add_geom(const XmlNode &n, int bodyid, const std::string &childclass) {
        AttrMap a = resolve(n, childclass);
        std::string ctx = "geom";
        if (const std::string *nm = aget(a, "name")) ctx += " " + *nm;
        int type = CM_GEOM_SPHERE;
        if (const std::string *t = aget(a, "type")) {
            static const char *names[] = {"plane", "hfield", "sphere", "capsule", "ellipsoid", "cylinder", "box", "mesh"};
            type = -1;
            for (int i = 0; i < 8; ++i)
                if (*t == names[i]) type = i;
            if (type < 0) return fail("unknown geom type " + *t);
        }
        double size[3] = {0, 0, 0}, pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0};
        if (const std::string *v = aget(a, "size"))
            if (!parse_doubles(*v, size, 1, 3)) return fail("bad size in " + ctx);
        if (!avec(a, "pos", pos, 3, ctx.c_str())) return false;
        if (!orientation(a, quat, ctx.c_str())) return false;
        if (const std::string *v = aget(a, "fromto")) {
            double ft[6];
            if (!parse_doubles(*v, ft, 6, 6)) return fail("bad fromto in " + ctx);
            double vec[3] = {ft[0] - ft[3], ft[1] - ft[4], ft[2] - ft[5]};
            double len = norm3(vec);
            if (type == CM_GEOM_CAPSULE || type == CM_GEOM_CYLINDER) size[1] = len / 2;
            else if (type == CM_GEOM_BOX || type == CM_GEOM_ELLIPSOID) { size[2] = len / 2; size[1] = size[0]; }
            else return fail("fromto on unsupported geom type in " + ctx);
            for (int i = 0; i < 3; ++i) pos[i] = 0.5 * (ft[i] + ft[i + 3]);
            z2quat(quat, vec);
        }
        int contype = 1, conaff = 1, condim = 3, priority = 0, group = 0;
        if (const std::string *v = aget(a, "contype")) contype = atoi(v->c_str());
        if (const std::string *v = aget(a, "conaffinity")) conaff = atoi(v->c_str());
        if (const std::string *v = aget(a, "condim")) condim = atoi(v->c_str());
        if (const std::string *v = aget(a, "priority")) priority = atoi(v->c_str());
        if (const std::string *v = aget(a, "group")) group = atoi(v->c_str());
        double friction[3] = {1, 0.005, 0.0001}, solref[2] = {0.02, 1}, solimp[5] = {0.9, 0.95, 0.001, 0.5, 2};
        double solmix = 1, margin = 0, gap = 0, rgba[4] = {0.5, 0.5, 0.5, 1}, mass = -1, density = 1000;
        if (const std::string *v = aget(a, "friction")) parse_doubles(*v, friction, 1, 3);
        if (const std::string *v = aget(a, "solref")) parse_doubles(*v, solref, 1, 2);
        if (const std::string *v = aget(a, "solimp")) parse_doubles(*v, solimp, 1, 5);
        if (const std::string *v = aget(a, "solmix")) solmix = atof(v->c_str());
        if (const std::string *v = aget(a, "margin")) margin = atof(v->c_str());
        if (const std::string *v = aget(a, "gap")) gap = atof(v->c_str());
        if (const std::string *v = aget(a, "rgba")) parse_doubles(*v, rgba, 4, 4);
        if (const std::string *v = aget(a, "mass")) mass = atof(v->c_str());
        if (const std::string *v = aget(a, "density")) density = atof(v->c_str());
        int dataid = -1;
        if (type == CM_GEOM_HFIELD) {
            const std::string *h = aget(a, "hfield");
            if (!h) return fail("hfield geom without hfield attribute");
            dataid = -1;
            for (size_t i = 0; i < m.hfield_name.size(); ++i)
                if (m.hfield_name[i] == *h) dataid = (int)i;
            if (dataid < 0) return fail("unknown hfield " + *h);
        }
        double rbound = 0;
        switch (type) {
            case CM_GEOM_SPHERE: rbound = size[0]; break;
            case CM_GEOM_CAPSULE: rbound = size[0] + size[1]; break;
            case CM_GEOM_CYLINDER: rbound = std::sqrt(size[0] * size[0] + size[1] * size[1]); break;
            case CM_GEOM_BOX: case CM_GEOM_ELLIPSOID: rbound = norm3(size); break;
            default: rbound = 0;
        }
        const std::string *nm = aget(a, "name");
        m.geom_name.push_back(nm ? *nm : std::string());
        m.geom_type.push_back(type);
        m.geom_bodyid.push_back(bodyid);
        m.geom_contype.push_back(contype);
        m.geom_conaffinity.push_back(conaff);
        m.geom_condim.push_back(condim);
        m.geom_priority.push_back(priority);
        m.geom_group.push_back(group);
        m.geom_dataid.push_back(dataid);
        m.geom_pos.insert(m.geom_pos.end(), pos, pos + 3);
        m.geom_quat.insert(m.geom_quat.end(), quat, quat + 4);
        m.geom_size.insert(m.geom_size.end(), size, size + 3);
        m.geom_friction.insert(m.geom_friction.end(), friction, friction + 3);
        m.geom_solref.insert(m.geom_solref.end(), solref, solref + 2);
        m.geom_solimp.insert(m.geom_solimp.end(), solimp, solimp + 5);
        m.geom_solmix.push_back(solmix);
        m.geom_margin.push_back(margin);
        m.geom_gap.push_back(gap);
        m.geom_rbound.push_back(rbound);
        for (int i = 0; i < 4; ++i) m.geom_rgba.push_back((float)rgba[i]);
        double user[8] = {0};
        if (const std::string *v = aget(a, "user")) parse_doubles(*v, user, 0, 8);
        for (int i = 0; i < m.nuser_geom; ++i) m.geom_user.push_back(user[i]);
        geom_massdens.emplace_back(mass, density);
        return true;
    }

    bool add_site(const XmlNode &n, int bodyid, const std::string &childclass) {
        AttrMap a = resolve(n, childclass);
        double pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0};
        if (!avec(a, "pos", pos, 3, "site")) return false;
        if (!orientation(a, quat, "site")) return false;
        if (const std::string *v = aget(a, "fromto")) {
            double ft[6];
            if (!parse_doubles(*v, ft, 6, 6)) return fail("bad site fromto");
            double vec[3] = {ft[0] - ft[3], ft[1] - ft[4], ft[2] - ft[5]};
            for (int i = 0; i < 3; ++i) pos[i] = 0.5 * (ft[i] + ft[i + 3]);
            z2quat(quat, vec);
        }
        const std::string *nm = aget(a, "name");
        m.site_name.push_back(nm ? *nm : std::string());
        m.site_bodyid.push_back(bodyid);
        m.site_pos.insert(m.site_pos.end(), pos, pos + 3);
        m.site_quat.insert(m.site_quat.end(), quat, quat + 4);
        return true;
    }

    bool add_joint(const XmlNode &n, int bodyid, const std::string &childclass) {
        bool freejoint = n.tag == "freejoint";
        AttrMap a = resolve(n, childclass, "joint");
        int type = CM_JNT_HINGE;
        if (freejoint) type = CM_JNT_FREE;
        else if (const std::string *t = aget(a, "type")) {
            if (*t == "free") type = CM_JNT_FREE;
            else if (*t == "ball") type = CM_JNT_BALL;
            else if (*t == "slide") type = CM_JNT_SLIDE;
            else if (*t == "hinge") type = CM_JNT_HINGE;
            else return fail("unknown joint type " + *t);
        }
        std::string ctx = "joint";
        const std::string *nm = aget(a, "name");
        if (nm) ctx += " " + *nm;
        double pos[3] = {0, 0, 0}, axis[3] = {0, 0, 1}, range[2] = {0, 0};
        if (!avec(a, "pos", pos, 3, ctx.c_str())) return false;
        if (!avec(a, "axis", axis, 3, ctx.c_str())) return false;
        normalize3(axis);
        if (!avec(a, "range", range, 2, ctx.c_str())) return false;
        double ref = 0, springref = 0, stiffness = 0, damping = 0, armature = 0, margin = 0;
        if (const std::string *v = aget(a, "ref")) ref = atof(v->c_str());
        if (const std::string *v = aget(a, "springref")) springref = atof(v->c_str());
        if (const std::string *v = aget(a, "stiffness")) stiffness = atof(v->c_str());
        if (const std::string *v = aget(a, "damping")) damping = atof(v->c_str());
        if (const std::string *v = aget(a, "armature")) armature = atof(v->c_str());
        if (const std::string *v = aget(a, "margin")) margin = atof(v->c_str());
        if (aget(a, "frictionloss") && atof(aget(a, "frictionloss")->c_str()) != 0)
            return fail("joint frictionloss not in the supported subset: " + ctx);
        bool limited = abool(a, "limited", false);
        if (type == CM_JNT_FREE) limited = false;
        if (limited && type == CM_JNT_BALL) return fail("limited ball joints not in the supported subset: " + ctx);
        double solref[2] = {0.02, 1}, solimp[5] = {0.9, 0.95, 0.001, 0.5, 2};
        if (const std::string *v = aget(a, "solreflimit")) parse_doubles(*v, solref, 1, 2);
        if (const std::string *v = aget(a, "solimplimit")) parse_doubles(*v, solimp, 1, 5);
        const double d2r = degree ? M_PI / 180.0 : 1.0;
        if (type == CM_JNT_HINGE) {
            range[0] *= d2r; range[1] *= d2r; ref *= d2r; springref *= d2r;
        }
        int nq = type == CM_JNT_FREE ? 7 : type == CM_JNT_BALL ? 4 : 1;
        int nv = type == CM_JNT_FREE ? 6 : type == CM_JNT_BALL ? 3 : 1;
        int jid = m.njnt++;
        m.jnt_name.push_back(nm ? *nm : std::string());
        m.jnt_type.push_back(type);
        m.jnt_qposadr.push_back(m.nq);
        m.jnt_dofadr.push_back(m.nv);
        m.jnt_bodyid.push_back(bodyid);
        m.jnt_limited.push_back(limited ? 1 : 0);
        m.jnt_pos.insert(m.jnt_pos.end(), pos, pos + 3);
        m.jnt_axis.insert(m.jnt_axis.end(), axis, axis + 3);
        m.jnt_range.insert(m.jnt_range.end(), range, range + 2);
        m.jnt_stiffness.push_back(stiffness);
        m.jnt_margin.push_back(margin);
        m.jnt_solref.insert(m.jnt_solref.end(), solref, solref + 2);
        m.jnt_solimp.insert(m.jnt_solimp.end(), solimp, solimp + 5);
        if (type == CM_JNT_FREE) {
            /* qpos0 of a free joint is the body's frame in its parent (= world) */
            for (int i = 0; i < 3; ++i) { m.qpos0.push_back(m.body_pos[3 * bodyid + i]); }
            for (int i = 0; i < 4; ++i) { m.qpos0.push_back(m.body_quat[4 * bodyid + i]); }
            for (int i = 0; i < 7; ++i) m.qpos_spring.push_back(m.qpos0[m.nq + i]);
        } else if (type == CM_JNT_BALL) {
            double q[4] = {1, 0, 0, 0};
            m.qpos0.insert(m.qpos0.end(), q, q + 4);
            m.qpos_spring.insert(m.qpos_spring.end(), q, q + 4);
        } else {
            m.qpos0.push_back(ref);
            m.qpos_spring.push_back(springref);
        }
        for (int k = 0; k < nv; ++k) {
            int parent;
            if (k > 0) parent = m.nv + k - 1;
            else {
                /* last dof of this body so far, else last dof of the nearest ancestor with dofs */
                parent = -1;
                if (m.body_dofnum[bodyid] > 0) parent = m.body_dofadr[bodyid] + m.body_dofnum[bodyid] - 1;
                else {
                    int b = m.body_parentid[bodyid];
                    while (b > 0 && m.body_dofnum[b] == 0) b = m.body_parentid[b];
                    if (b > 0) parent = m.body_dofadr[b] + m.body_dofnum[b] - 1;
                }
            }
            m.dof_bodyid.push_back(bodyid);
            m.dof_jntid.push_back(jid);
            m.dof_parentid.push_back(parent);
            m.dof_armature.push_back(armature);
            m.dof_damping.push_back(damping);
            m.dof_invweight0.push_back(0);
        }
        if (m.body_jntnum[bodyid] == 0) { m.body_jntadr[bodyid] = jid; m.body_dofadr[bodyid] = m.nv; }
        m.body_jntnum[bodyid]++;
        m.body_dofnum[bodyid] += nv;
        m.nq += nq;
        m.nv += nv;
        return true;
    }

    /* inertia of a body without <inertial>: sum over its geoms (inertiafromgeom='auto') */
    bool inertia_from_geoms(int bodyid, int g0, int g1) {
        double M = 0, com[3] = {0, 0, 0};
        std::vector<double> gm(g1 - g0, 0.0);
        for (int g = g0; g < g1; ++g) {
            const double *s = &m.geom_size[3 * g];
            double vol = 0;
            switch (m.geom_type[g]) {
                case CM_GEOM_SPHERE: vol = 4.0 / 3.0 * M_PI * s[0] * s[0] * s[0]; break;
                case CM_GEOM_CAPSULE: vol = M_PI * s[0] * s[0] * (2 * s[1]) + 4.0 / 3.0 * M_PI * s[0] * s[0] * s[0]; break;
                case CM_GEOM_CYLINDER: vol = M_PI * s[0] * s[0] * 2 * s[1]; break;
                case CM_GEOM_BOX: vol = 8 * s[0] * s[1] * s[2]; break;
                case CM_GEOM_ELLIPSOID: vol = 4.0 / 3.0 * M_PI * s[0] * s[1] * s[2]; break;
                default: vol = 0;
            }
            double mass = geom_massdens[g].first >= 0 ? geom_massdens[g].first : geom_massdens[g].second * vol;
            gm[g - g0] = mass;
            M += mass;
            for (int i = 0; i < 3; ++i) com[i] += mass * m.geom_pos[3 * g + i];
        }
        if (M <= 0) return true; /* massless static body */
        for (int i = 0; i < 3; ++i) com[i] /= M;
        double I[9] = {0};
        for (int g = g0; g < g1; ++g) {
            double mass = gm[g - g0];
            if (mass <= 0) continue;
            const double *s = &m.geom_size[3 * g];
            double d[3] = {0, 0, 0};
            switch (m.geom_type[g]) {
                case CM_GEOM_SPHERE: d[0] = d[1] = d[2] = 0.4 * mass * s[0] * s[0]; break;
                case CM_GEOM_BOX:
                    d[0] = mass / 3 * (s[1] * s[1] + s[2] * s[2]);
                    d[1] = mass / 3 * (s[0] * s[0] + s[2] * s[2]);
                    d[2] = mass / 3 * (s[0] * s[0] + s[1] * s[1]);
                    break;
                case CM_GEOM_CYLINDER:
                    d[0] = d[1] = mass * (3 * s[0] * s[0] + 4 * s[1] * s[1]) / 12;
                    d[2] = mass * s[0] * s[0] / 2;
                    break;
                case CM_GEOM_ELLIPSOID:
                    d[0] = mass / 5 * (s[1] * s[1] + s[2] * s[2]);
                    d[1] = mass / 5 * (s[0] * s[0] + s[2] * s[2]);
                    d[2] = mass / 5 * (s[0] * s[0] + s[1] * s[1]);
                    break;
                case CM_GEOM_CAPSULE: {
                    double r = s[0], h = 2 * s[1];
                    double vc = M_PI * r * r * h, vs = 4.0 / 3.0 * M_PI * r * r * r;
                    double mc = mass * vc / (vc + vs), ms = mass * vs / (vc + vs);
                    d[2] = mc * r * r / 2 + ms * 0.4 * r * r;
                    d[0] = d[1] = mc * (3 * r * r + h * h) / 12 + ms * (0.4 * r * r + h * h / 4 + 3.0 / 8 * r * h);
                } break;
                default: return fail("cannot infer inertia from this geom type");
            }
            double R[9];
            quat2mat(R, &m.geom_quat[4 * g]);
            double off[3];
            for (int i = 0; i < 3; ++i) off[i] = m.geom_pos[3 * g + i] - com[i];
            double o2 = dot3(off, off);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double v = 0;
                    for (int k = 0; k < 3; ++k) v += R[3 * i + k] * d[k] * R[3 * j + k];
                    v += mass * ((i == j ? o2 : 0.0) - off[i] * off[j]);
                    I[3 * i + j] += v;
                }
        }
        double ev[3], q[4];
        eig3(I, ev, q);
        m.body_mass[bodyid] = M;
        for (int i = 0; i < 3; ++i) { m.body_ipos[3 * bodyid + i] = com[i]; m.body_inertia[3 * bodyid + i] = ev[i]; }
        for (int i = 0; i < 4; ++i) m.body_iquat[4 * bodyid + i] = q[i];
        return true;
    }

    int new_body(const std::string &name, int parent, const double *pos, const double *quat) {
        int id = m.nbody++;
        m.body_name.push_back(name);
        m.body_parentid.push_back(parent);
        m.body_jntadr.push_back(-1); m.body_jntnum.push_back(0);
        m.body_dofadr.push_back(-1); m.body_dofnum.push_back(0);
        m.body_geomadr.push_back(-1); m.body_geomnum.push_back(0);
        m.body_pos.insert(m.body_pos.end(), pos, pos + 3);
        m.body_quat.insert(m.body_quat.end(), quat, quat + 4);
        double z3[3] = {0, 0, 0}, q1[4] = {1, 0, 0, 0}, z2[2] = {0, 0};
        m.body_ipos.insert(m.body_ipos.end(), z3, z3 + 3);
        m.body_iquat.insert(m.body_iquat.end(), q1, q1 + 4);
        m.body_mass.push_back(0);
        m.body_inertia.insert(m.body_inertia.end(), z3, z3 + 3);
        m.body_invweight0.insert(m.body_invweight0.end(), z2, z2 + 2);
        m.body_subtreemass.push_back(0);
        return id;
    }

    /* MuJoCo numbers bodies depth-first, but numbers joints/geoms/sites body by body:
     * a body's own elements are added when the body is visited, before its children. */
    bool body_contents(const XmlNode &n, int bodyid, const std::string &childclass) {
        bool has_inertial = false;
        for (auto &k : n.kids) {
            if (k->tag == "inertial") {
                has_inertial = true;
                AttrMap a;
                for (auto &at : k->attr) a[at.first] = at.second;
                double pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0}, mass = 0;
                if (!avec(a, "pos", pos, 3, "inertial")) return false;
                if (!orientation(a, quat, "inertial")) return false;
                if (const std::string *v = aget(a, "mass")) mass = atof(v->c_str());
                double inertia[3] = {0, 0, 0}, iq[4] = {quat[0], quat[1], quat[2], quat[3]};
                if (const std::string *v = aget(a, "fullinertia")) {
                    double f[6];
                    if (!parse_doubles(*v, f, 6, 6)) return fail("bad fullinertia");
                    double I[9] = {f[0], f[3], f[4], f[3], f[1], f[5], f[4], f[5], f[2]};
                    double q2[4];
                    eig3(I, inertia, q2);
                    mulquat(iq, quat, q2);
                } else if (const std::string *v = aget(a, "diaginertia")) {
                    if (!parse_doubles(*v, inertia, 3, 3)) return fail("bad diaginertia");
                } else
                    return fail("inertial without fullinertia/diaginertia");
                m.body_mass[bodyid] = mass;
                for (int i = 0; i < 3; ++i) { m.body_ipos[3 * bodyid + i] = pos[i]; m.body_inertia[3 * bodyid + i] = inertia[i]; }
                for (int i = 0; i < 4; ++i) m.body_iquat[4 * bodyid + i] = iq[i];
            }
        }
        for (auto &k : n.kids)
            if (k->tag == "joint" || k->tag == "freejoint")
                if (!add_joint(*k, bodyid, childclass)) return false;
        int g0 = (int)m.geom_type.size();
        for (auto &k : n.kids)
            if (k->tag == "geom")
                if (!add_geom(*k, bodyid, childclass)) return false;
        int g1 = (int)m.geom_type.size();
        m.body_geomadr[bodyid] = g1 > g0 ? g0 : -1;
        m.body_geomnum[bodyid] = g1 - g0;
        for (auto &k : n.kids)
            if (k->tag == "site")
                if (!add_site(*k, bodyid, childclass)) return false;
        for (auto &k : n.kids)
            if (k->tag == "camera") {
                const std::string *nm = k->get("name");
                m.cam_name.push_back(nm ? *nm : std::string());
                m.ncam++;
            }
        if (bodyid > 0 && !has_inertial && inertiafromgeom_auto)
            if (!inertia_from_geoms(bodyid, g0, g1)) return false;
        for (auto &k : n.kids) {
            const std::string &t = k->tag;
            if (t != "inertial" && t != "joint" && t != "freejoint" && t != "geom" && t != "site" && t != "camera" &&
                t != "light" && t != "body")
                return fail("unsupported element <" + t + "> inside body");
        }
        return true;
    }

    bool body_tree(const XmlNode &n, int bodyid, const std::string &childclass) {
        for (auto &k : n.kids) {
            if (k->tag != "body") continue;
            std::string cc = childclass;
            if (const std::string *c = k->get("childclass")) cc = *c;
            AttrMap a;
            for (auto &at : k->attr) a[at.first] = at.second;
            double pos[3] = {0, 0, 0}, quat[4];
            if (!avec(a, "pos", pos, 3, "body")) return false;
            if (!orientation(a, quat, "body")) return false;
            const std::string *nm = k->get("name");
            int id = new_body(nm ? *nm : std::string(), bodyid, pos, quat);
            if (!body_contents(*k, id, cc)) return false;
            if (!body_tree(*k, id, cc)) return false;
        }
        return true;
    }
};

/* -------------------------------------------- host kinematics at a pose --- */
struct HostKin {
    const HostModel &m;
    std::vector<double> xpos, xquat, xmat, xipos, ximat, xanchor, xaxis, cdof, com;
    explicit HostKin(const HostModel &hm) : m(hm) {}
    void run(const double *qpos) {
        int nb = m.nbody;
        xpos.assign(3 * nb, 0); xquat.assign(4 * nb, 0); xmat.assign(9 * nb, 0);
        xipos.assign(3 * nb, 0); ximat.assign(9 * nb, 0);
        xanchor.assign(3 * m.njnt, 0); xaxis.assign(3 * m.njnt, 0);
        xquat[0] = 1; xmat[0] = xmat[4] = xmat[8] = 1; ximat[0] = ximat[4] = ximat[8] = 1;
        for (int b = 1; b < nb; ++b) {
            int p = m.body_parentid[b];
            double pos[3], quat[4];
            bool isfree = m.body_jntnum[b] == 1 && m.jnt_type[m.body_jntadr[b]] == CM_JNT_FREE;
            if (isfree) {
                int qa = m.jnt_qposadr[m.body_jntadr[b]];
                for (int i = 0; i < 3; ++i) pos[i] = qpos[qa + i];
                for (int i = 0; i < 4; ++i) quat[i] = qpos[qa + 3 + i];
                normalize4(quat);
                int j = m.body_jntadr[b];
                for (int i = 0; i < 3; ++i) { xanchor[3 * j + i] = pos[i]; xaxis[3 * j + i] = i == 2; }
            } else {
                rotvec(pos, &xmat[9 * p], &m.body_pos[3 * b]);
                for (int i = 0; i < 3; ++i) pos[i] += xpos[3 * p + i];
                mulquat(quat, &xquat[4 * p], &m.body_quat[4 * b]);
                for (int jj = 0; jj < m.body_jntnum[b]; ++jj) {
                    int j = m.body_jntadr[b] + jj, qa = m.jnt_qposadr[j];
                    double anchor[3], axis[3];
                    rotvecquat(anchor, &m.jnt_pos[3 * j], quat);
                    for (int i = 0; i < 3; ++i) anchor[i] += pos[i];
                    rotvecquat(axis, &m.jnt_axis[3 * j], quat);
                    for (int i = 0; i < 3; ++i) { xanchor[3 * j + i] = anchor[i]; xaxis[3 * j + i] = axis[i]; }
                    if (m.jnt_type[j] == CM_JNT_SLIDE) {
                        for (int i = 0; i < 3; ++i) pos[i] += axis[i] * (qpos[qa] - m.qpos0[qa]);
                    } else {
                        double ql[4];
                        if (m.jnt_type[j] == CM_JNT_BALL) {
                            for (int i = 0; i < 4; ++i) ql[i] = qpos[qa + i];
                            normalize4(ql);
                        } else {
                            double ang = qpos[qa] - m.qpos0[qa];
                            ql[0] = std::cos(ang / 2);
                            for (int i = 0; i < 3; ++i) ql[1 + i] = m.jnt_axis[3 * j + i] * std::sin(ang / 2);
                        }
                        mulquat(quat, quat, ql);
                        double r[3];
                        rotvecquat(r, &m.jnt_pos[3 * j], quat);
                        for (int i = 0; i < 3; ++i) pos[i] = anchor[i] - r[i];
                    }
                }
            }
            normalize4(quat);
            for (int i = 0; i < 3; ++i) xpos[3 * b + i] = pos[i];
            for (int i = 0; i < 4; ++i) xquat[4 * b + i] = quat[i];
            quat2mat(&xmat[9 * b], quat);
            rotvec(&xipos[3 * b], &xmat[9 * b], &m.body_ipos[3 * b]);
            for (int i = 0; i < 3; ++i) xipos[3 * b + i] += pos[i];
            double qi[4];
            mulquat(qi, quat, &m.body_iquat[4 * b]);
            quat2mat(&ximat[9 * b], qi);
        }
    }
    /* translational (jacp) and rotational (jacr) Jacobian of a world point attached to body b; 3 x nv each */
    void jac(int b, const double *point, std::vector<double> &jacp, std::vector<double> &jacr) const {
        int nv = m.nv;
        jacp.assign(3 * nv, 0); jacr.assign(3 * nv, 0);
        while (b > 0) {
            for (int jj = m.body_jntnum[b] - 1; jj >= 0; --jj) {
                int j = m.body_jntadr[b] + jj, d = m.jnt_dofadr[j];
                const double *an = &xanchor[3 * j], *ax = &xaxis[3 * j];
                double off[3] = {point[0] - an[0], point[1] - an[1], point[2] - an[2]};
                auto rotdof = [&](int dd, const double *axis) {
                    double c[3];
                    cross3(c, axis, off);
                    for (int i = 0; i < 3; ++i) { jacp[i * nv + dd] = c[i]; jacr[i * nv + dd] = axis[i]; }
                };
                switch (m.jnt_type[j]) {
                    case CM_JNT_SLIDE:
                        for (int i = 0; i < 3; ++i) jacp[i * nv + d] = ax[i];
                        break;
                    case CM_JNT_HINGE: rotdof(d, ax); break;
                    case CM_JNT_BALL:
                        for (int k = 0; k < 3; ++k) {
                            double a[3] = {xmat[9 * b + k], xmat[9 * b + 3 + k], xmat[9 * b + 6 + k]};
                            rotdof(d + k, a);
                        }
                        break;
                    case CM_JNT_FREE:
                        for (int k = 0; k < 3; ++k) jacp[k * nv + d + k] = 1;
                        for (int k = 0; k < 3; ++k) {
                            double a[3] = {xmat[9 * b + k], xmat[9 * b + 3 + k], xmat[9 * b + 6 + k]};
                            rotdof(d + 3 + k, a);
                        }
                        break;
                }
            }
            b = m.body_parentid[b];
        }
    }
};

/* dense symmetric positive definite solve helpers (model-compile time only) */
bool cholesky(std::vector<double> &A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (d <= 0) return false;
        d = std::sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return true;
}
void chol_solve(const std::vector<double> &L, int n, double *x) {
    for (int i = 0; i < n; ++i) {
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * x[k];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}

}  // namespace

/* ===================================================== HostModel methods === */
int HostModel::name2id(int objtype, const char *name) const {
    const std::vector<std::string> *v = nullptr;
    switch (objtype) {
        case OBJ_BODY: case 2 /* mjOBJ_XBODY */: v = &body_name; break;
        case OBJ_JOINT: v = &jnt_name; break;
        case OBJ_GEOM: v = &geom_name; break;
        case OBJ_SITE: v = &site_name; break;
        case OBJ_CAMERA: v = &cam_name; break;
        case OBJ_HFIELD: v = &hfield_name; break;
        case OBJ_EQUALITY: v = &eq_name; break;
        case OBJ_ACTUATOR: v = &act_name; break;
        case OBJ_SENSOR: v = &sensor_name; break;
        default: return -1;
    }
    if (!name || !*name) return -1;
    for (size_t i = 0; i < v->size(); ++i)
        if ((*v)[i] == name) return (int)i;
    return -1;
}

const char *HostModel::id2name(int objtype, int id) const {
    const std::vector<std::string> *v = nullptr;
    switch (objtype) {
        case OBJ_BODY: case 2: v = &body_name; break;
        case OBJ_JOINT: v = &jnt_name; break;
        case OBJ_GEOM: v = &geom_name; break;
        case OBJ_SITE: v = &site_name; break;
        case OBJ_CAMERA: v = &cam_name; break;
        case OBJ_HFIELD: v = &hfield_name; break;
        case OBJ_EQUALITY: v = &eq_name; break;
        case OBJ_ACTUATOR: v = &act_name; break;
        case OBJ_SENSOR: v = &sensor_name; break;
        default: return nullptr;
    }
    if (id < 0 || id >= (int)v->size() || (*v)[id].empty()) return nullptr;
    return (*v)[id].c_str();
}

void HostModel::set_const() {
    /* subtree masses */
    for (int b = 0; b < nbody; ++b) body_subtreemass[b] = body_mass[b];
    for (int b = nbody - 1; b > 0; --b) body_subtreemass[body_parentid[b]] += body_subtreemass[b];

    HostKin kin(*this);
    kin.run(qpos0.data());
    /* dense M(qpos0) = sum_b Jb^T diag(m, I_b) Jb + armature */
    std::vector<double> M((size_t)nv * nv, 0.0), jp, jr;
    for (int b = 1; b < nbody; ++b) {
        if (body_mass[b] <= 0) continue;
        kin.jac(b, &kin.xipos[3 * b], jp, jr);
        const double *R = &kin.ximat[9 * b];
        /* rotate jr into the inertial frame: w_local = R^T jr */
        std::vector<double> jl(3 * nv);
        for (int d = 0; d < nv; ++d)
            for (int i = 0; i < 3; ++i)
                jl[i * nv + d] = R[0 + i] * jr[d] + R[3 + i] * jr[nv + d] + R[6 + i] * jr[2 * nv + d];
        for (int r = 0; r < nv; ++r)
            for (int c = 0; c < nv; ++c) {
                double s = 0;
                for (int i = 0; i < 3; ++i)
                    s += body_mass[b] * jp[i * nv + r] * jp[i * nv + c] +
                         body_inertia[3 * b + i] * jl[i * nv + r] * jl[i * nv + c];
                M[r * nv + c] += s;
            }
    }
    double mean = 0;
    for (int d = 0; d < nv; ++d) { M[d * nv + d] += dof_armature[d]; mean += M[d * nv + d]; }
    meaninertia = nv > 0 ? mean / nv : 1.0;
    std::vector<double> L = M;
    if (nv == 0 || !cholesky(L, nv)) return;

    /* static weld ids: a jointless body rides on its parent's weld */
    std::vector<int> weld(nbody, 0);
    for (int b = 1; b < nbody; ++b) weld[b] = body_jntnum[b] > 0 ? b : weld[body_parentid[b]];

    for (int b = 0; b < nbody; ++b) {
        body_invweight0[2 * b] = body_invweight0[2 * b + 1] = 0;
        if (b == 0 || weld[b] == 0) continue;
        kin.jac(b, &kin.xipos[3 * b], jp, jr);
        double tr = 0, rot = 0;
        std::vector<double> col(nv);
        for (int i = 0; i < 3; ++i) {
            for (int d = 0; d < nv; ++d) col[d] = jp[i * nv + d];
            chol_solve(L, nv, col.data());
            for (int d = 0; d < nv; ++d) tr += jp[i * nv + d] * col[d];
            for (int d = 0; d < nv; ++d) col[d] = jr[i * nv + d];
            chol_solve(L, nv, col.data());
            for (int d = 0; d < nv; ++d) rot += jr[i * nv + d] * col[d];
        }
        body_invweight0[2 * b] = tr / 3;
        body_invweight0[2 * b + 1] = rot / 3;
    }
    /* dof_invweight0 = diag(M^-1), averaged over the dofs of ball / free joints */
    std::vector<double> dinv(nv);
    for (int d = 0; d < nv; ++d) {
        std::vector<double> e(nv, 0.0);
        e[d] = 1;
        chol_solve(L, nv, e.data());
        dinv[d] = e[d];
    }
    for (int j = 0; j < njnt; ++j) {
        int d = jnt_dofadr[j];
        switch (jnt_type[j]) {
            case CM_JNT_FREE: {
                double a = (dinv[d] + dinv[d + 1] + dinv[d + 2]) / 3, r = (dinv[d + 3] + dinv[d + 4] + dinv[d + 5]) / 3;
                for (int k = 0; k < 3; ++k) { dof_invweight0[d + k] = a; dof_invweight0[d + 3 + k] = r; }
            } break;
            case CM_JNT_BALL: {
                double r = (dinv[d] + dinv[d + 1] + dinv[d + 2]) / 3;
                for (int k = 0; k < 3; ++k) dof_invweight0[d + k] = r;
            } break;
            default: dof_invweight0[d] = dinv[d];
        }
    }
}

bool HostModel::compile(cm_model_t *o, std::string *err) const {
    auto fail = [&](const std::string &s) {
        if (err) *err = s;
        return false;
    };
    memset(o, 0, sizeof *o);
    if (nbody > CM_MAXBODY || njnt > CM_MAXJNT || nq > CM_MAXQ || nv > CM_MAXV || nu > CM_MAXU || neq > CM_MAXEQ ||
        nsite > CM_MAXSITE || nsensor > CM_MAXSENSOR || nsensordata > CM_MAXSENSORDATA)
        return fail("model exceeds the compiled capacity limits (cm_model.h)");
    o->nq = nq; o->nv = nv; o->nu = nu; o->nbody = nbody; o->njnt = njnt; o->neq = neq; o->nsite = nsite;
    o->nsensor = nsensor; o->nsensordata = nsensordata;
    /* (the 127-row solve's cross-wave turn word counts two per sweep in 12 bits, pk_wide_solve.h) */
    if (iterations > 2000) return fail("option iterations > 2000 is not supported (the in-scope models ask for 50)");
    o->iterations = iterations; o->flags = flags;
    o->timestep = timestep; o->tolerance = tolerance; o->meaninertia = meaninertia;
    for (int i = 0; i < 3; ++i) { o->gravity[i] = gravity[i]; o->magnetic[i] = magnetic[i]; }
    o->hfield_geom = -1;
    o->hfield_nrow = hfield_nrow; o->hfield_ncol = hfield_ncol;
    for (int i = 0; i < 4; ++i) o->hfield_size[i] = hfield_size[i];

    std::vector<int> weld(nbody, 0);
    for (int b = 0; b < nbody; ++b) {
        o->body_parentid[b] = body_parentid[b];
        o->body_jntadr[b] = body_jntadr[b]; o->body_jntnum[b] = body_jntnum[b];
        o->body_dofadr[b] = body_dofadr[b]; o->body_dofnum[b] = body_dofnum[b];
        o->body_depth[b] = b == 0 ? 0 : o->body_depth[body_parentid[b]] + 1;
        if (o->body_depth[b] > o->maxdepth) o->maxdepth = o->body_depth[b];
        if (b > 0) weld[b] = body_jntnum[b] > 0 ? b : weld[body_parentid[b]];
        o->body_weldid[b] = weld[b];
        o->body_rootid[b] = b == 0 ? 0 : (body_parentid[b] == 0 ? b : o->body_rootid[body_parentid[b]]);
        for (int i = 0; i < 3; ++i) {
            o->body_pos[b][i] = body_pos[3 * b + i]; o->body_ipos[b][i] = body_ipos[3 * b + i];
            o->body_inertia[b][i] = body_inertia[3 * b + i];
        }
        for (int i = 0; i < 4; ++i) { o->body_quat[b][i] = body_quat[4 * b + i]; o->body_iquat[b][i] = body_iquat[4 * b + i]; }
        o->body_mass[b] = body_mass[b];
        quat2mat(o->body_mat[b], &body_quat[4 * b]);
        quat2mat(o->body_imat[b], &body_iquat[4 * b]);
        o->body_invweight0[b][0] = body_invweight0[2 * b]; o->body_invweight0[b][1] = body_invweight0[2 * b + 1];
    }
    /* subtree ranges (ids are depth-first) and dof masks */
    for (int b = 0; b < nbody; ++b) {
        int e = b + 1;
        while (e < nbody) {
            int a = e;
            bool inside = false;
            while (a > 0) { if (a == b) { inside = true; break; } a = body_parentid[a]; }
            if (b == 0) inside = true;
            if (!inside) break;
            ++e;
        }
        o->body_subtreeend[b] = e;
        uint64_t mask = 0;
        for (int a = b; a > 0; a = body_parentid[a])
            for (int k = 0; k < body_dofnum[a]; ++k) mask |= 1ull << (body_dofadr[a] + k);
        o->body_dofmask[b] = mask;
    }
    for (int j = 0; j < njnt; ++j) {
        o->jnt_type[j] = jnt_type[j]; o->jnt_qposadr[j] = jnt_qposadr[j]; o->jnt_dofadr[j] = jnt_dofadr[j];
        o->jnt_bodyid[j] = jnt_bodyid[j]; o->jnt_limited[j] = jnt_limited[j];
        o->jnt_parentbody[j] = jnt_type[j] == CM_JNT_FREE ? -1 : body_parentid[jnt_bodyid[j]];
        o->jnt_ref[j] = qpos0[jnt_qposadr[j]];
        o->jnt_liminvweight[j] = dof_invweight0[jnt_dofadr[j]];
        for (int i = 0; i < 3; ++i) { o->jnt_pos[j][i] = jnt_pos[3 * j + i]; o->jnt_axis[j][i] = jnt_axis[3 * j + i]; }
        for (int i = 0; i < 2; ++i) { o->jnt_range[j][i] = jnt_range[2 * j + i]; o->jnt_solref[j][i] = jnt_solref[2 * j + i]; }
        for (int i = 0; i < 5; ++i) o->jnt_solimp[j][i] = jnt_solimp[5 * j + i];
        o->jnt_stiffness[j] = jnt_stiffness[j]; o->jnt_margin[j] = jnt_margin[j];
    }
    /* kinematics records: leading slides + one rotational joint per body (cm_model.h cm_kinrec_t) */
    o->kin_simple = 1;
    for (int b = 0; b < nbody; ++b) {
        cm_kinrec_t &kr = o->body_kin[b];
        memset(&kr, 0, sizeof kr);
        const int j0 = body_jntadr[b], jn = body_jntnum[b];
        int nsl = 0;
        while (nsl < jn && jnt_type[j0 + nsl] == CM_JNT_SLIDE) ++nsl;
        const int rest = jn - nsl;
        const int rj = rest >= 1 ? j0 + nsl : -1;
        const bool freebody = rj >= 0 && jnt_type[rj] == CM_JNT_FREE;
        if (nsl > CM_MAXSLIDE || rest > 1 || (freebody && nsl > 0)) { o->kin_simple = 0; nsl = nsl > CM_MAXSLIDE ? CM_MAXSLIDE : nsl; }
        kr.nslide = nsl; kr.jnt0 = jn > 0 ? j0 : 0; kr.rot_jnt = rj; kr.rot_type = rj >= 0 ? jnt_type[rj] : -1;
        const double ident[4] = {1, 0, 0, 0};
        for (int i = 0; i < 4; ++i) kr.quat[i] = freebody ? ident[i] : body_quat[4 * b + i];
        quat2mat(kr.mat, kr.quat);
        for (int i = 0; i < 3; ++i) kr.pos[i] = body_pos[3 * b + i];
        for (int sl = 0; sl < nsl; ++sl) {
            const int j = j0 + sl;
            kr.slide_qadr[sl] = jnt_qposadr[j]; kr.slide_ref[sl] = qpos0[jnt_qposadr[j]];
            rotvec(kr.slide_axis_p[sl], kr.mat, &jnt_axis[3 * j]);
            rotvec(kr.slide_pos_p[sl], kr.mat, &jnt_pos[3 * j]);
        }
        if (rj >= 0) {
            kr.rot_qadr = jnt_qposadr[rj]; kr.rot_ref = qpos0[jnt_qposadr[rj]];
            if (freebody) kr.rot_axis[2] = kr.rot_axis_p[2] = 1.0;
            else {
                for (int i = 0; i < 3; ++i) { kr.rot_axis[i] = jnt_axis[3 * rj + i]; kr.rot_pos[i] = jnt_pos[3 * rj + i]; }
                rotvec(kr.rot_axis_p, kr.mat, kr.rot_axis);
                rotvec(kr.rot_pos_p, kr.mat, kr.rot_pos);
            }
        }
    }
    for (int i = 0; i < nq; ++i) { o->qpos0[i] = qpos0[i]; o->qpos_spring[i] = qpos_spring[i]; }
    for (int d = 0; d < nv; ++d) {
        o->dof_bodyid[d] = dof_bodyid[d]; o->dof_jntid[d] = dof_jntid[d]; o->dof_parentid[d] = dof_parentid[d];
        o->dof_armature[d] = dof_armature[d]; o->dof_damping[d] = dof_damping[d]; o->dof_invweight0[d] = dof_invweight0[d];
    }
    if (o->maxdepth > 27) return fail("kinematic tree deeper than 27 levels");
    for (int b = 0; b < nbody; ++b)
        for (int r = 0, unit = 1; r < 3; ++r, unit *= 3)
            for (int i = 1; i <= 2; ++i) {
                int x = b;
                for (int t = 0; t < i * unit; ++t) x = x > 0 ? body_parentid[x] : 0;
                o->body_anc3[b][2 * r + i - 1] = x;
            }
    o->nroot = 0;
    for (int b = 1; b < nbody; ++b) {
        o->body_nchild[b] = 0;
        if (body_parentid[b] == 0 && weld[b] != 0) {
            if (o->nroot >= 4) return fail("more than 4 kinematic trees");
            o->root_body[o->nroot++] = b;
        }
    }
    o->body_nchild[0] = 0;
    for (int b = 1; b < nbody; ++b) {
        int p = body_parentid[b];
        if (p == 0) continue;
        if (o->body_nchild[p] >= 4) return fail("a body has more than 4 children");
        o->body_child[p][o->body_nchild[p]++] = b;
    }
    for (int d = 0; d < nv; ++d) {
        uint64_t anc = 0, vel = 0;
        for (int a = dof_parentid[d]; a >= 0; a = dof_parentid[a]) {
            anc |= 1ull << a;
            if (dof_jntid[a] != dof_jntid[d]) vel |= 1ull << a;
        }
        o->dof_ancmask[d] = anc;
        o->dof_velmask[d] = vel;
    }
    for (int d = 0; d < nv; ++d) {
        int depth = 0;
        for (int a = dof_parentid[d]; a >= 0; a = dof_parentid[a]) ++depth;
        if (depth >= 32) return fail("dof chain deeper than 32");
        for (int r = 0, unit = 1; r < 3; ++r, unit *= 4)
            for (int i = 1; i <= 3; ++i) {
                int a = d;
                for (int t = 0; t < i * unit && a >= 0; ++t) a = dof_parentid[a];
                o->dof_anc4[d][3 * r + i - 1] = a;
            }
        int src = dof_parentid[d];
        while (src >= 0 && dof_jntid[src] == dof_jntid[d]) src = dof_parentid[src];
        const int j = dof_jntid[d];
        if (jnt_type[j] == CM_JNT_FREE) src = (d - jnt_dofadr[j] >= 3) ? jnt_dofadr[j] + 2 : -1;
        o->dof_vinsrc[d] = src;
    }
    for (int b = 0; b < nbody; ++b) {
        int a = b;
        while (a > 0 && body_dofnum[a] == 0) a = body_parentid[a];
        o->body_lastdof[b] = a > 0 ? body_dofadr[a] + body_dofnum[a] - 1 : -1;
    }
    for (int d = 0; d < nv; ++d) {
        uint64_t desc = 1ull << d;
        for (int k = 0; k < nv; ++k)
            if ((o->dof_ancmask[k] >> d) & 1ull) desc |= 1ull << k;
        o->dof_descmask[d] = desc;
    }
    /* the kernels implement the impedance sigmoid for exponents 1 and 2 only (all in-scope models use 2) */
    auto power_ok = [](double p) { return p == 1.0 || p == 2.0; };
    for (int j = 0; j < njnt; ++j) if (!power_ok(jnt_solimp[5 * j + 4])) return fail("solimp power other than 1 or 2 is not supported");
    for (int e = 0; e < neq; ++e) if (!power_ok(eq_solimp[5 * e + 4])) return fail("solimp power other than 1 or 2 is not supported");
    for (int g = 0; g < ngeom; ++g)
        if ((geom_contype[g] || geom_conaffinity[g]) && !power_ok(geom_solimp[5 * g + 4])) return fail("solimp power other than 1 or 2 is not supported");
    /* collision geoms */
    std::vector<int> cg;
    for (int g = 0; g < ngeom; ++g)
        if (geom_contype[g] || geom_conaffinity[g]) cg.push_back(g);
    if ((int)cg.size() > CM_MAXGEOM) return fail("too many collision geoms");
    o->ngeom = (int)cg.size();
    for (int k = 0; k < o->ngeom; ++k) {
        int g = cg[k];
        o->geom_type[k] = geom_type[g]; o->geom_bodyid[k] = geom_bodyid[g]; o->geom_condim[k] = geom_condim[g];
        o->geom_priority[k] = geom_priority[g]; o->geom_contype[k] = geom_contype[g];
        o->geom_conaffinity[k] = geom_conaffinity[g]; o->geom_fullid[k] = g;
        for (int i = 0; i < 3; ++i) {
            o->geom_pos[k][i] = geom_pos[3 * g + i]; o->geom_size[k][i] = geom_size[3 * g + i];
            o->geom_friction[k][i] = geom_friction[3 * g + i];
        }
        for (int i = 0; i < 4; ++i) o->geom_quat[k][i] = geom_quat[4 * g + i];
        quat2mat(o->geom_mat[k], &geom_quat[4 * g]);
        for (int i = 0; i < 2; ++i) o->geom_solref[k][i] = geom_solref[2 * g + i];
        for (int i = 0; i < 5; ++i) o->geom_solimp[k][i] = geom_solimp[5 * g + i];
        o->geom_solmix[k] = geom_solmix[g]; o->geom_margin[k] = geom_margin[g]; o->geom_gap[k] = geom_gap[g];
        o->geom_rbound[k] = geom_rbound[g];
        if (geom_type[g] == CM_GEOM_HFIELD) o->hfield_geom = k;
        o->geom_farstatic[k] = (weld[geom_bodyid[g]] == 0 && geom_type[g] != CM_GEOM_PLANE && geom_type[g] != CM_GEOM_HFIELD) ? 1 : 0;
        if (geom_type[g] == CM_GEOM_MESH || geom_type[g] == CM_GEOM_ELLIPSOID || geom_type[g] == CM_GEOM_CYLINDER)
            return fail("collision geom type not in the supported subset (mesh/ellipsoid/cylinder)");
    }
    /* candidate pairs */
    struct P { int b1, b2, g1, g2; };
    std::vector<P> pairs;
    for (int a = 0; a < o->ngeom; ++a)
        for (int b = a + 1; b < o->ngeom; ++b) {
            int ba = o->geom_bodyid[a], bb = o->geom_bodyid[b];
            int wa = weld[ba], wb = weld[bb];
            if (wa == wb) continue;
            int wpa = weld[body_parentid[wa]], wpb = weld[body_parentid[wb]];
            if (wa != 0 && wb != 0 && (wa == wpb || wb == wpa)) continue;
            bool ok = (o->geom_contype[a] & o->geom_conaffinity[b]) || (o->geom_contype[b] & o->geom_conaffinity[a]);
            if (!ok) continue;
            P p;
            p.g1 = a; p.g2 = b;
            if (o->geom_type[a] > o->geom_type[b]) std::swap(p.g1, p.g2);
            p.b1 = std::min(ba, bb); p.b2 = std::max(ba, bb);
            pairs.push_back(p);
        }
    std::stable_sort(pairs.begin(), pairs.end(), [](const P &x, const P &y) {
        if (x.b1 != y.b1) return x.b1 < y.b1;
        return x.b2 < y.b2;
    });
    auto multi = [&](const P &x) {
        int t1 = o->geom_type[x.g1], t2 = o->geom_type[x.g2];
        return (t1 == CM_GEOM_PLANE && t2 == CM_GEOM_BOX) || (t1 == CM_GEOM_BOX && t2 == CM_GEOM_BOX);
    };
    /* static geoms other than planes / height fields can be culled as a block when the robot is nowhere near them */
    auto far_static = [&](int g) {
        return weld[o->geom_bodyid[g]] == 0 && o->geom_type[g] != CM_GEOM_PLANE && o->geom_type[g] != CM_GEOM_HFIELD;
    };
    auto cullable = [&](const P &x) { return !multi(x) && (far_static(x.g1) || far_static(x.g2)); };
    std::stable_partition(pairs.begin(), pairs.end(), [&](const P &x) { return !multi(x); });
    o->npair_simple = 0;
    for (auto &x : pairs) if (!multi(x)) o->npair_simple++;
    std::stable_partition(pairs.begin(), pairs.begin() + o->npair_simple, [&](const P &x) { return !cullable(x); });
    o->npair_always = 0;
    for (int i = 0; i < o->npair_simple; ++i) if (!cullable(pairs[i])) o->npair_always++;
    /* reach of every kinematic tree around its root body's origin */
    for (int b = 0; b < nbody; ++b) o->body_reach[b] = 0;
    for (int k = 0; k < o->ngeom; ++k) {
        int b = o->geom_bodyid[k];
        if (weld[b] == 0) continue;
        double len = std::sqrt(o->geom_pos[k][0] * o->geom_pos[k][0] + o->geom_pos[k][1] * o->geom_pos[k][1] + o->geom_pos[k][2] * o->geom_pos[k][2]) + o->geom_rbound[k];
        int r = o->body_rootid[b];
        for (int a = b; a != r && a > 0; a = body_parentid[a]) {
            len += std::sqrt(body_pos[3 * a] * body_pos[3 * a] + body_pos[3 * a + 1] * body_pos[3 * a + 1] + body_pos[3 * a + 2] * body_pos[3 * a + 2]);
            for (int jj = 0; jj < body_jntnum[a]; ++jj) {
                int j = body_jntadr[a] + jj;
                len += 2 * std::sqrt(jnt_pos[3 * j] * jnt_pos[3 * j] + jnt_pos[3 * j + 1] * jnt_pos[3 * j + 1] + jnt_pos[3 * j + 2] * jnt_pos[3 * j + 2]);
                if (jnt_type[j] == CM_JNT_SLIDE || jnt_type[j] == CM_JNT_FREE) len = 1e30; /* unbounded travel below the root */
            }
        }
        /* joints of the root itself: rotations about an offset anchor move the origin by at most 2 |jnt_pos| */
        for (int jj = 0; jj < body_jntnum[r]; ++jj) {
            int j = body_jntadr[r] + jj;
            len += 2 * std::sqrt(jnt_pos[3 * j] * jnt_pos[3 * j] + jnt_pos[3 * j + 1] * jnt_pos[3 * j + 1] + jnt_pos[3 * j + 2] * jnt_pos[3 * j + 2]);
        }
        if (len > o->body_reach[r]) o->body_reach[r] = len;
    }
    if ((int)pairs.size() > CM_MAXPAIR) return fail("too many candidate collision pairs");
    o->npair = (int)pairs.size();
    o->nhfpair = 0;
    for (int i = 0; i < CM_MAXHFPAIR; ++i) o->hfpair[i] = 0;
    for (int i = 0; i < CM_MAXPAIR; ++i) o->pair_hfslot[i] = -1;
    for (int i = 0; i < o->npair; ++i) {
        const int g1 = pairs[i].g1, g2 = pairs[i].g2;
        o->pair_geom1[i] = g1; o->pair_geom2[i] = g2;
        o->pair_type[i] = o->geom_type[g1] | (o->geom_type[g2] << 8);
        if (o->geom_type[g1] == CM_GEOM_HFIELD && i < o->npair_simple &&
            (o->geom_type[g2] == CM_GEOM_SPHERE || o->geom_type[g2] == CM_GEOM_CAPSULE)) {
            if (o->nhfpair >= CM_MAXHFPAIR) return fail("too many height-field collision pairs (CM_MAXHFPAIR)");
            o->pair_hfslot[i] = o->nhfpair; o->hfpair[o->nhfpair] = i;
            o->nhfpair++;
        }
        o->pair_margin[i] = std::max(o->geom_margin[g1], o->geom_margin[g2]);
        o->pair_includemargin[i] = o->pair_margin[i] - std::max(o->geom_gap[g1], o->geom_gap[g2]);
        o->pair_rbound[i][0] = o->geom_rbound[g1]; o->pair_rbound[i][1] = o->geom_rbound[g2];
        for (int k = 0; k < 3; ++k) { o->pair_size[i][k] = o->geom_size[g1][k]; o->pair_size[i][3 + k] = o->geom_size[g2][k]; }
        {
            const int bb[2] = {o->geom_bodyid[g1], o->geom_bodyid[g2]};
            o->pair_invweight[i] = 0;
            for (int k = 0; k < 2; ++k) {
                o->pair_root[i][k] = o->body_rootid[bb[k]];
                o->pair_dofmask[i][k] = bb[k] > 0 ? o->body_dofmask[bb[k]] : 0ull;
                o->pair_invweight[i] += o->body_invweight0[bb[k]][0];
            }
        }
        /* contact parameter mixing (MuJoCo mj_contactParam): the geom with the higher priority wins outright; at equal
         * priority condim and friction take the maximum and solref / solimp are blended by solmix */
        const int pa = o->geom_priority[g1], pb = o->geom_priority[g2];
        if (pa != pb) {
            const int g = pa > pb ? g1 : g2;
            o->pair_condim[i] = o->geom_condim[g];
            for (int k = 0; k < 2; ++k) o->pair_solref[i][k] = o->geom_solref[g][k];
            for (int k = 0; k < 5; ++k) o->pair_solimp[i][k] = o->geom_solimp[g][k];
            for (int k = 0; k < 3; ++k) o->pair_friction[i][k] = o->geom_friction[g][k];
        } else {
            o->pair_condim[i] = std::max(o->geom_condim[g1], o->geom_condim[g2]);
            const double s1 = o->geom_solmix[g1], s2 = o->geom_solmix[g2];
            double mix;
            if (s1 >= CM_MINVAL && s2 >= CM_MINVAL) mix = s1 / (s1 + s2);
            else if (s1 < CM_MINVAL && s2 < CM_MINVAL) mix = 0.5;
            else mix = s1 < CM_MINVAL ? 0.0 : 1.0;
            if (o->geom_solref[g1][0] > 0 && o->geom_solref[g2][0] > 0)
                for (int k = 0; k < 2; ++k) o->pair_solref[i][k] = mix * o->geom_solref[g1][k] + (1 - mix) * o->geom_solref[g2][k];
            else
                for (int k = 0; k < 2; ++k) o->pair_solref[i][k] = std::min(o->geom_solref[g1][k], o->geom_solref[g2][k]);
            for (int k = 0; k < 5; ++k) o->pair_solimp[i][k] = mix * o->geom_solimp[g1][k] + (1 - mix) * o->geom_solimp[g2][k];
            for (int k = 0; k < 3; ++k) o->pair_friction[i][k] = std::max(o->geom_friction[g1][k], o->geom_friction[g2][k]);
        }
    }

    for (int e = 0; e < neq; ++e) {
        o->eq_body1[e] = eq_body1[e]; o->eq_body2[e] = eq_body2[e]; o->eq_active[e] = eq_active[e];
        for (int i = 0; i < 6; ++i) o->eq_data[e][i] = eq_data[6 * e + i];
        for (int i = 0; i < 2; ++i) o->eq_solref[e][i] = eq_solref[2 * e + i];
        for (int i = 0; i < 5; ++i) o->eq_solimp[e][i] = eq_solimp[5 * e + i];
        {
            const int bb[2] = {eq_body1[e], eq_body2[e]};
            o->eq_invweight[e] = 0;
            for (int k = 0; k < 2; ++k) {
                o->eq_root[e][k] = o->body_rootid[bb[k]];
                o->eq_dofmask[e][k] = o->body_dofmask[bb[k]];
                o->eq_invweight[e] += o->body_invweight0[bb[k]][0];
            }
        }
    }
    for (int u = 0; u < nu; ++u) {
        int j = act_jntid[u];
        o->act_dofid[u] = jnt_dofadr[j]; o->act_qposadr[u] = jnt_qposadr[j];
        o->act_ctrllimited[u] = act_ctrllimited[u];
        o->act_gear[u] = act_gear[6 * u];
        o->act_maxrpm[u] = nuser_actuator > 0 ? act_user[(size_t)nuser_actuator * u] : 0.0;
        o->act_ctrlrange[u][0] = act_ctrlrange[2 * u]; o->act_ctrlrange[u][1] = act_ctrlrange[2 * u + 1];
    }
    for (int d = 0; d < nv; ++d) {
        const int j = dof_jntid[d];
        const bool scalar = jnt_type[j] == CM_JNT_HINGE || jnt_type[j] == CM_JNT_SLIDE;
        o->dof_qadr[d] = jnt_qposadr[j];
        o->dof_stiffness[d] = scalar ? jnt_stiffness[j] : 0.0;
        o->dof_springref[d] = scalar ? qpos_spring[jnt_qposadr[j]] : 0.0;
        o->dof_act[d] = 0; o->dof_gear[d] = 0.0; o->dof_ctrl_lo[d] = -1e300; o->dof_ctrl_hi[d] = 1e300;
        for (int u = 0; u < nu; ++u) {
            if (o->act_dofid[u] != d) continue;
            o->dof_act[d] = u; o->dof_gear[d] = o->act_gear[u];
            if (o->act_ctrllimited[u]) { o->dof_ctrl_lo[d] = o->act_ctrlrange[u][0]; o->dof_ctrl_hi[d] = o->act_ctrlrange[u][1]; }
        }
    }
    for (int s = 0; s < nsite; ++s) {
        o->site_bodyid[s] = site_bodyid[s];
        for (int i = 0; i < 3; ++i) o->site_pos[s][i] = site_pos[3 * s + i];
        for (int i = 0; i < 4; ++i) o->site_quat[s][i] = site_quat[4 * s + i];
    }
    for (int s = 0; s < nsensor; ++s) {
        o->sensor_type[s] = sensor_type[s]; o->sensor_objid[s] = sensor_objid[s];
        o->sensor_adr[s] = sensor_adr[s]; o->sensor_dim[s] = sensor_dim[s]; o->sensor_cutoff[s] = sensor_cutoff[s];
        o->sensor_qadr[s] = -1; o->sensor_gain[s] = 1.0; o->sensor_body[s] = 0; o->sensor_root[s] = 0;
        for (int i = 0; i < 4; ++i) o->sensor_squat[s][i] = i == 0 ? 1.0 : 0.0;
        for (int i = 0; i < 3; ++i) o->sensor_spos[s][i] = 0.0;
        if (sensor_type[s] == CM_SENS_ACTUATORPOS) { o->sensor_qadr[s] = o->act_qposadr[sensor_objid[s]]; o->sensor_gain[s] = o->act_gear[sensor_objid[s]]; }
        else if (sensor_type[s] == CM_SENS_JOINTPOS) o->sensor_qadr[s] = jnt_qposadr[sensor_objid[s]];
        else if (sensor_type[s] >= CM_SENS_FRAMEQUAT && sensor_type[s] <= CM_SENS_MAGNETOMETER) {
            const int site = sensor_objid[s];
            o->sensor_body[s] = site_bodyid[site];
            o->sensor_root[s] = o->body_rootid[site_bodyid[site]];
            for (int i = 0; i < 4; ++i) o->sensor_squat[s][i] = site_quat[4 * site + i];
            for (int i = 0; i < 3; ++i) o->sensor_spos[s][i] = site_pos[3 * site + i];
        }
        o->sensor_bits[s] = nuser_sensor > 0 ? (int)sensor_user[(size_t)nuser_sensor * s] : 0;
        o->sensor_slot[s] = -1;
        if (sensor_type[s] == CM_SENS_ACCELEROMETER) {
            int slot = 0;
            for (int t = 0; t < s; ++t) if (sensor_type[t] == CM_SENS_ACCELEROMETER) ++slot;
            if (slot > 1) return fail("more than two accelerometers");
            o->sensor_slot[s] = slot;
        }
    }
    /* what an env-step of this model may use of the contact list and of the constraint rows (cm_model.h: CM_MAXCON / CM_MAXEFC).  The
     * caps follow the contact definition: 16 contacts / 63 rows (never reached by the shipped definitions: 0 of 163 840 stress
     * windows, profiles/round4) -- and 32 / 127 with CM_FLAG_HFPRISM, whose contact sets need them; the 127-row instantiation of the
     * step kernel exists for the 32-dof Cassie dof tree.  (Round 5 first gave every model on that tree the wide caps: the 127-row
     * pass then sits behind every launch, and its workgroups -- two empty SIMDs, 84 KB of LDS -- wait for the other env range's
     * kernel to drain even when their list is empty: -4 % on config 2, profiles/round5/wide_pass_ab.txt.) */
    bool cassie32 = nv == ck::TopoCassie32::nv && o->kin_simple && o->maxdepth <= ck::TopoCassie32::body_levels;
    for (int k = 0; cassie32 && k < nv; ++k) cassie32 = o->dof_ancmask[k] == ck::TopoCassie32::table[k];
    const bool wide_caps = cassie32 && (o->flags & CM_FLAG_HFPRISM) != 0 && o->nhfpair > 0;
    o->maxcon = wide_caps ? CM_MAXCON : CM_MAXCON_NARROW;
    o->maxefc = wide_caps ? CM_MAXEFC : CM_MAXEFC_NARROW;

    /* kinematics at qpos0 for the device's set_const kernel (cm_model_t::body_xpos0 ...): the very values set_const() above works
     * with, per dof what HostKin::jac makes of the dof's joint */
    {
        HostKin kin(*this);
        kin.run(qpos0.data());
        for (int b = 0; b < nbody; ++b) {
            for (int i = 0; i < 3; ++i) o->body_xpos0[b][i] = kin.xpos[3 * b + i];
            for (int i = 0; i < 9; ++i) { o->body_xmat0[b][i] = kin.xmat[9 * b + i]; o->body_ximat0[b][i] = kin.ximat[9 * b + i]; }
        }
        for (int j = 0; j < njnt; ++j) {
            const int b = jnt_bodyid[j], d = jnt_dofadr[j];
            const double *an = &kin.xanchor[3 * j], *ax = &kin.xaxis[3 * j];
            auto put = [&](int dd, int trans, const double *axis) {
                o->dof_trans0[dd] = trans;
                for (int i = 0; i < 3; ++i) { o->dof_axis0[dd][i] = axis[i]; o->dof_anchor0[dd][i] = trans ? 0.0 : an[i]; }
            };
            switch (jnt_type[j]) {
                case CM_JNT_SLIDE: put(d, 1, ax); break;
                case CM_JNT_HINGE: put(d, 0, ax); break;
                case CM_JNT_BALL:
                case CM_JNT_FREE: {
                    const int r0 = jnt_type[j] == CM_JNT_FREE ? d + 3 : d;
                    if (jnt_type[j] == CM_JNT_FREE)
                        for (int k = 0; k < 3; ++k) { const double e[3] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0}; put(d + k, 1, e); }
                    for (int k = 0; k < 3; ++k) {
                        const double a[3] = {kin.xmat[9 * b + k], kin.xmat[9 * b + 3 + k], kin.xmat[9 * b + 6 + k]};
                        put(r0 + k, 0, a);
                    }
                } break;
            }
        }
    }
    /* the model's own parameter block (cm_envparams_t): what the step kernel reads where an env has no block of its own */
    {
        cm_envparams_t *p = &o->params;
        for (int b = 0; b < nbody; ++b) {
            p->body_mass[b] = o->body_mass[b];
            for (int i = 0; i < 3; ++i) { p->body_ipos[b][i] = o->body_ipos[b][i]; p->body_inertia[b][i] = o->body_inertia[b][i]; }
            for (int i = 0; i < 2; ++i) p->body_invweight0[b][i] = o->body_invweight0[b][i];
        }
        for (int d = 0; d < nv; ++d) { p->dof_damping[d] = o->dof_damping[d]; p->dof_invweight0[d] = o->dof_invweight0[d]; }
        for (int g = 0; g < o->ngeom; ++g) for (int i = 0; i < 3; ++i) p->geom_friction[g][i] = o->geom_friction[g][i];
        p->meaninertia = o->meaninertia;
        for (int j = 0; j < njnt; ++j) p->jnt_liminvweight[j] = o->jnt_liminvweight[j];
        for (int e = 0; e < neq; ++e) p->eq_invweight[e] = o->eq_invweight[e];
        for (int i = 0; i < o->npair; ++i) {
            p->pair_invweight[i] = o->pair_invweight[i];
            for (int k = 0; k < 3; ++k) p->pair_friction[i][k] = o->pair_friction[i][k];
        }
    }
    return true;
}

/* ================================================================ loader === */
bool load_mjcf(const std::string &path, HostModel *out, std::string *errp) {
    std::string dummy;
    std::string &err = errp ? *errp : dummy;
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    std::string src = ss.str();
    XmlParser xp(src);
    std::unique_ptr<XmlNode> root = xp.parse();
    if (!root) { err = "XML parse error: " + xp.err; return false; }
    if (root->tag != "mujoco") { err = "root element is not <mujoco>"; return false; }

    *out = HostModel();
    HostModel &m = *out;
    Compiler C(m, err);

    auto section = [&](const char *tag) -> const XmlNode * {
        for (auto &k : root->kids)
            if (k->tag == tag) return k.get();
        return nullptr;
    };
    if (const XmlNode *c = section("compiler")) {
        if (const std::string *v = c->get("angle")) C.degree = *v != "radian";
        if (const std::string *v = c->get("inertiafromgeom")) C.inertiafromgeom_auto = *v != "false";
        if (const std::string *v = c->get("coordinate"))
            if (*v != "local") { err = "global coordinates not supported"; return false; }
    }
    if (const XmlNode *s = section("size")) {
        if (const std::string *v = s->get("nuser_sensor")) m.nuser_sensor = atoi(v->c_str());
        if (const std::string *v = s->get("nuser_actuator")) m.nuser_actuator = atoi(v->c_str());
        if (const std::string *v = s->get("nuser_geom")) m.nuser_geom = atoi(v->c_str());
    }
    if (const XmlNode *o = section("option")) {
        for (auto &a : o->attr) {
            if (a.first == "timestep") m.timestep = atof(a.second.c_str());
            else if (a.first == "iterations") m.iterations = atoi(a.second.c_str());
            else if (a.first == "tolerance") m.tolerance = atof(a.second.c_str());
            else if (a.first == "gravity") parse_doubles(a.second, m.gravity, 3, 3);
            else if (a.first == "magnetic") parse_doubles(a.second, m.magnetic, 3, 3);
            else if (a.first == "solver") {
                if (a.second != "PGS") { err = "only solver='PGS' is supported (all in-scope models use it)"; return false; }
                m.solver_pgs = 1;
            } else if (a.first == "integrator") {
                if (a.second != "Euler") { err = "only the Euler integrator is supported"; return false; }
            } else if (a.first == "cone") {
                if (a.second != "pyramidal") { err = "only pyramidal friction cones are supported"; return false; }
            } else { err = "unsupported <option> attribute " + a.first; return false; }
        }
        for (auto &k : o->kids) { err = "unsupported <option> child <" + k->tag + ">"; return false; }
    }
    if (!m.solver_pgs) { err = "model does not select solver='PGS'; Newton/CG are outside the supported subset"; return false; }
    if (const XmlNode *v = section("visual"))
        for (auto &k : v->kids)
            if (k->tag == "map")
                if (const std::string *z = k->get("znear")) m.vis_znear = (float)atof(z->c_str());
    if (const XmlNode *d = section("default")) C.read_defaults(*d, "main", "");
    if (const XmlNode *a = section("asset"))
        for (auto &k : a->kids)
            if (k->tag == "hfield") {
                if (m.nhfield) { err = "more than one hfield asset"; return false; }
                if (k->get("file")) { err = "hfield from file not supported (set samples at run time)"; return false; }
                const std::string *nm = k->get("name"), *nr = k->get("nrow"), *nc = k->get("ncol"), *sz = k->get("size");
                if (!nr || !nc || !sz || !parse_doubles(*sz, m.hfield_size, 4, 4)) { err = "bad hfield asset"; return false; }
                m.hfield_name.push_back(nm ? *nm : std::string());
                m.hfield_nrow = atoi(nr->c_str()); m.hfield_ncol = atoi(nc->c_str());
                m.nhfield = 1; m.nhfielddata = m.hfield_nrow * m.hfield_ncol;
                m.hfield_data.assign(m.nhfielddata, 0.f);
            }
    for (auto &k : root->kids) {
        const std::string &t = k->tag;
        if (t == "tendon" || t == "contact" || t == "keyframe" || t == "custom" || t == "include") {
            err = "unsupported MJCF section <" + t + ">";
            return false;
        }
    }

    /* world body */
    const XmlNode *wb = section("worldbody");
    if (!wb) { err = "no <worldbody>"; return false; }
    double z3[3] = {0, 0, 0}, q1[4] = {1, 0, 0, 0};
    C.new_body("world", 0, z3, q1);
    if (!C.body_contents(*wb, 0, "")) return false;
    if (!C.body_tree(*wb, 0, "")) return false;
    m.ngeom = (int)m.geom_type.size();
    m.nsite = (int)m.site_bodyid.size();

    /* equality */
    if (const XmlNode *eq = section("equality"))
        for (auto &k : eq->kids) {
            if (k->tag != "connect") { err = "unsupported equality <" + k->tag + ">"; return false; }
            AttrMap a = C.resolve(*k, "", "equality");
            const std::string *b1 = Compiler::aget(a, "body1"), *b2 = Compiler::aget(a, "body2");
            if (!b1) { err = "connect without body1"; return false; }
            int i1 = m.name2id(OBJ_BODY, b1->c_str()), i2 = b2 ? m.name2id(OBJ_BODY, b2->c_str()) : 0;
            if (i1 < 0 || i2 < 0) { err = "connect references unknown body"; return false; }
            double anchor[3] = {0, 0, 0}, solref[2] = {0.02, 1}, solimp[5] = {0.9, 0.95, 0.001, 0.5, 2};
            if (!C.avec(a, "anchor", anchor, 3, "connect")) return false;
            if (const std::string *v = Compiler::aget(a, "solref")) parse_doubles(*v, solref, 1, 2);
            if (const std::string *v = Compiler::aget(a, "solimp")) parse_doubles(*v, solimp, 1, 5);
            const std::string *nm = Compiler::aget(a, "name");
            m.eq_name.push_back(nm ? *nm : std::string());
            m.eq_body1.push_back(i1); m.eq_body2.push_back(i2);
            m.eq_active.push_back(Compiler::abool(a, "active", true) ? 1 : 0);
            double data[6] = {anchor[0], anchor[1], anchor[2], 0, 0, 0};
            m.eq_data.insert(m.eq_data.end(), data, data + 6);
            m.eq_solref.insert(m.eq_solref.end(), solref, solref + 2);
            m.eq_solimp.insert(m.eq_solimp.end(), solimp, solimp + 5);
            m.neq++;
        }
    /* second anchors: same world point at qpos0, expressed in body2 */
    {
        HostKin kin(m);
        kin.run(m.qpos0.data());
        for (int e = 0; e < m.neq; ++e) {
            int b1 = m.eq_body1[e], b2 = m.eq_body2[e];
            double w[3], l[3];
            rotvec(w, &kin.xmat[9 * b1], &m.eq_data[6 * e]);
            for (int i = 0; i < 3; ++i) w[i] += kin.xpos[3 * b1 + i] - kin.xpos[3 * b2 + i];
            rotvecT(l, &kin.xmat[9 * b2], w);
            for (int i = 0; i < 3; ++i) m.eq_data[6 * e + 3 + i] = l[i];
        }
    }

    /* actuators */
    if (const XmlNode *ac = section("actuator"))
        for (auto &k : ac->kids) {
            if (k->tag != "motor") { err = "unsupported actuator <" + k->tag + ">"; return false; }
            AttrMap a = C.resolve(*k, "");
            const std::string *jn = Compiler::aget(a, "joint");
            if (!jn) { err = "motor without joint transmission"; return false; }
            int j = m.name2id(OBJ_JOINT, jn->c_str());
            if (j < 0 || (m.jnt_type[j] != CM_JNT_HINGE && m.jnt_type[j] != CM_JNT_SLIDE)) { err = "motor joint must be an existing hinge/slide"; return false; }
            double gear[6] = {1, 0, 0, 0, 0, 0}, range[2] = {0, 0}, user[8] = {0};
            if (const std::string *v = Compiler::aget(a, "gear")) parse_doubles(*v, gear, 1, 6);
            if (const std::string *v = Compiler::aget(a, "ctrlrange")) parse_doubles(*v, range, 2, 2);
            if (const std::string *v = Compiler::aget(a, "user")) parse_doubles(*v, user, 0, 8);
            if (Compiler::aget(a, "forcelimited") && Compiler::abool(a, "forcelimited", false)) { err = "forcelimited motors not supported"; return false; }
            const std::string *nm = Compiler::aget(a, "name");
            m.act_name.push_back(nm ? *nm : std::string());
            m.act_jntid.push_back(j);
            m.act_ctrllimited.push_back(Compiler::abool(a, "ctrllimited", false) ? 1 : 0);
            m.act_gear.insert(m.act_gear.end(), gear, gear + 6);
            m.act_ctrlrange.insert(m.act_ctrlrange.end(), range, range + 2);
            for (int i = 0; i < m.nuser_actuator; ++i) m.act_user.push_back(user[i]);
            m.nu++;
        }

    /* sensors */
    if (const XmlNode *se = section("sensor"))
        for (auto &k : se->kids) {
            int type = -1, dim = 0, objid = -1;
            const std::string &t = k->tag;
            auto ref = [&](const char *attr, int ot) -> int {
                const std::string *v = k->get(attr);
                return v ? m.name2id(ot, v->c_str()) : -1;
            };
            if (t == "actuatorpos") { type = CM_SENS_ACTUATORPOS; dim = 1; objid = ref("actuator", OBJ_ACTUATOR); }
            else if (t == "jointpos") {
                type = CM_SENS_JOINTPOS; dim = 1; objid = ref("joint", OBJ_JOINT);
                if (objid >= 0 && m.jnt_type[objid] != CM_JNT_HINGE && m.jnt_type[objid] != CM_JNT_SLIDE) objid = -1;
            }
            else if (t == "framequat") {
                type = CM_SENS_FRAMEQUAT; dim = 4;
                const std::string *ot = k->get("objtype");
                if (!ot || *ot != "site") { err = "framequat sensor only supported on sites"; return false; }
                objid = ref("objname", OBJ_SITE);
            }
            else if (t == "gyro") { type = CM_SENS_GYRO; dim = 3; objid = ref("site", OBJ_SITE); }
            else if (t == "accelerometer") { type = CM_SENS_ACCELEROMETER; dim = 3; objid = ref("site", OBJ_SITE); }
            else if (t == "magnetometer") { type = CM_SENS_MAGNETOMETER; dim = 3; objid = ref("site", OBJ_SITE); }
            else { err = "unsupported sensor <" + t + ">"; return false; }
            if (objid < 0) { err = "sensor <" + t + "> references an unknown object"; return false; }
            const std::string *nm = k->get("name");
            m.sensor_name.push_back(nm ? *nm : std::string());
            m.sensor_type.push_back(type); m.sensor_objid.push_back(objid);
            m.sensor_adr.push_back(m.nsensordata); m.sensor_dim.push_back(dim);
            double cutoff = 0, noise = 0, user[8] = {0};
            if (const std::string *v = k->get("cutoff")) cutoff = atof(v->c_str());
            if (const std::string *v = k->get("noise")) noise = atof(v->c_str());
            if (const std::string *v = k->get("user")) parse_doubles(*v, user, 0, 8);
            m.sensor_cutoff.push_back(cutoff); m.sensor_noise.push_back(noise);
            for (int i = 0; i < m.nuser_sensor; ++i) m.sensor_user.push_back(user[i]);
            m.nsensordata += dim;
            m.nsensor++;
        }

    m.set_const();
    return true;
}

/* ======================================================= serialisation === */
namespace {
template <class T> void put(std::ostream &o, const char *key, const std::vector<T> &v) {
    o << key << ' ' << v.size();
    char buf[40];
    for (auto &x : v) {
        snprintf(buf, sizeof buf, " %.17g", (double)x);
        o << buf;
    }
    o << '\n';
}
void puts_(std::ostream &o, const char *key, const std::vector<std::string> &v) {
    o << key << ' ' << v.size();
    for (auto &x : v) o << ' ' << (x.empty() ? std::string("~") : x);
    o << '\n';
}
}  // namespace

#define CM_SCALAR_FIELDS(X)                                                                                      \
    X(nq) X(nv) X(nu) X(nbody) X(njnt) X(ngeom) X(nsite) X(ncam) X(nsensor) X(nsensordata) X(neq) X(nhfield)         \
    X(nhfielddata) X(nuser_sensor) X(nuser_actuator) X(nuser_geom) X(timestep) X(tolerance) X(impratio)            \
    X(iterations) X(solver_pgs) X(flags) X(meaninertia) X(stat_extent) X(vis_znear) X(vis_zfar) X(hfield_nrow)     \
    X(hfield_ncol)
#define CM_VECTOR_FIELDS(X)                                                                                      \
    X(body_parentid) X(body_jntadr) X(body_jntnum) X(body_dofadr) X(body_dofnum) X(body_geomadr) X(body_geomnum)   \
    X(body_pos) X(body_quat) X(body_ipos) X(body_iquat) X(body_mass) X(body_inertia) X(body_invweight0)            \
    X(body_subtreemass) X(jnt_type) X(jnt_qposadr) X(jnt_dofadr) X(jnt_bodyid) X(jnt_limited) X(jnt_pos)           \
    X(jnt_axis) X(jnt_range) X(jnt_stiffness) X(jnt_margin) X(jnt_solref) X(jnt_solimp) X(qpos0) X(qpos_spring)    \
    X(dof_bodyid) X(dof_jntid) X(dof_parentid) X(dof_armature) X(dof_damping) X(dof_invweight0) X(geom_type)       \
    X(geom_bodyid) X(geom_contype) X(geom_conaffinity) X(geom_condim) X(geom_priority) X(geom_group)               \
    X(geom_dataid) X(geom_pos) X(geom_quat) X(geom_size) X(geom_friction) X(geom_solref) X(geom_solimp)            \
    X(geom_solmix) X(geom_margin) X(geom_gap) X(geom_rbound) X(geom_user) X(geom_rgba) X(site_bodyid) X(site_pos)  \
    X(site_quat) X(eq_body1) X(eq_body2) X(eq_active) X(eq_data) X(eq_solref) X(eq_solimp) X(act_jntid)            \
    X(act_ctrllimited) X(act_gear) X(act_ctrlrange) X(act_user) X(sensor_type) X(sensor_objid) X(sensor_adr)       \
    X(sensor_dim) X(sensor_cutoff) X(sensor_noise) X(sensor_user)
#define CM_NAME_FIELDS(X)                                                                                        \
    X(body_name) X(jnt_name) X(geom_name) X(site_name) X(act_name) X(sensor_name) X(eq_name) X(hfield_name)        \
    X(cam_name)

bool HostModel::save(const std::string &path) const {
    std::ofstream o(path.c_str());
    if (!o) return false;
    o << "cmodel 1\n";
    char buf[64];
#define X(f) snprintf(buf, sizeof buf, "%.17g", (double)f); o << #f << " 1 " << buf << '\n';
    CM_SCALAR_FIELDS(X)
#undef X
    std::vector<double> g(gravity, gravity + 3), mg(magnetic, magnetic + 3), hs(hfield_size, hfield_size + 4),
        sc(stat_center, stat_center + 3);
    put(o, "gravity", g); put(o, "magnetic", mg); put(o, "hfield_size", hs); put(o, "stat_center", sc);
#define X(f) put(o, #f, f);
    CM_VECTOR_FIELDS(X)
#undef X
#define X(f) puts_(o, #f, f);
    CM_NAME_FIELDS(X)
#undef X
    return (bool)o;
}

bool HostModel::load(const std::string &path, std::string *err) {
    std::ifstream in(path.c_str());
    if (!in) { if (err) *err = "cannot open " + path; return false; }
    *this = HostModel();
    std::string magic;
    int ver = 0;
    in >> magic >> ver;
    if (magic != "cmodel" || ver != 1) { if (err) *err = path + ": not a cmodel v1 file"; return false; }
    std::string key;
    size_t n;
    while (in >> key >> n) {
        bool done = false;
#define X(f) if (!done && key == #f) { double v; in >> v; f = (decltype(f))v; done = true; }
        CM_SCALAR_FIELDS(X)
#undef X
#define X(f) if (!done && key == #f) { f.resize(n); for (size_t i = 0; i < n; ++i) { double v; in >> v; f[i] = (decltype(f)::value_type)v; } done = true; }
        CM_VECTOR_FIELDS(X)
#undef X
#define X(f) if (!done && key == #f) { f.resize(n); for (size_t i = 0; i < n; ++i) { in >> f[i]; if (f[i] == "~") f[i].clear(); } done = true; }
        CM_NAME_FIELDS(X)
#undef X
        if (!done) {
            double *dst = nullptr;
            if (key == "gravity") dst = gravity;
            else if (key == "magnetic") dst = magnetic;
            else if (key == "hfield_size") dst = hfield_size;
            else if (key == "stat_center") dst = stat_center;
            if (!dst) { if (err) *err = path + ": unknown key " + key; return false; }
            for (size_t i = 0; i < n; ++i) in >> dst[i];
        }
        if (!in) { if (err) *err = path + ": truncated at key " + key; return false; }
    }
    hfield_data.assign(nhfielddata, 0.f);
    return true;
}

bool load_model_file(const std::string &path, HostModel *out, std::string *err) {
    size_t n = path.size();
    if (n >= 4 && path.compare(n - 4, 4, ".xml") == 0) return load_mjcf(path, out, err);
    return out->load(path, err);
}

}  // namespace cm
