// keyfile.cpp -- the text key-file format of ImageRepresentation::SaveRegions / LoadRegions
// (imagerepresentation.cpp:2139-2215) with saveAR / saveKP / loadAR / loadKP (:35-38, :89-99, :130-147).
//
//   <number of detectors>
//   <detector name> <number of descriptor classes>          per detector
//   <descriptor name> <number of regions>                   per class
//   <descriptor length>                                     only when the class has regions
//   id img_id img_reproj_id parent_id  KP(det_kp) KP(reproj_kp)  <len> d0 d1 ...      one line per region
//   KP = x y a11 a12 a21 a22 pyramid_scale octave_number s sub_type
//
// Numbers go through the same libstdc++ ostream << double / float / int as in the reference (6 significant
// digits), so a file written here is byte-identical to the one SaveRegions writes for the same lists.
#include <algorithm>
#include <fstream>
#include <sstream>
#include <string>
#include <string.h>
#include <vector>
#include "engine_api.hpp"

namespace mx {

static void save_kp(const modsx_keypoint &k, std::ostream &s) {   // saveKP, :35-38
  s << k.x << " " << k.y << " " << k.a11 << " " << k.a12 << " " << k.a21 << " " << k.a22 << " ";
  s << k.pyramid_scale << " " << k.octave_number << " " << k.s << " " << k.sub_type << " ";
}
static void load_kp(modsx_keypoint &k, std::istream &s) {         // loadKP, :130-132
  s >> k.x >> k.y >> k.a11 >> k.a12 >> k.a21 >> k.a22 >> k.pyramid_scale >> k.octave_number >> k.s >> k.sub_type;
}

int save_regions(const char *path, const modsx_region_class *classes, int nclasses) {
  std::ofstream f(path);
  if (!f.is_open()) { set_error(std::string("cannot open ") + path + " to save keypoints"); return MODSX_ERR_ARG; }
  // RegionVectorMap is a std::map<detector, std::map<descriptor, list>>: both levels come out sorted by name
  std::vector<std::string> dets;
  for (int i = 0; i < nclasses; i++) {
    bool seen = false;
    for (auto &d : dets) seen = seen || d == classes[i].det_name;
    if (!seen) dets.push_back(classes[i].det_name);
  }
  std::sort(dets.begin(), dets.end());
  f << dets.size() << std::endl;
  for (auto &det : dets) {
    std::vector<int> idx;
    for (int i = 0; i < nclasses; i++) if (det == classes[i].det_name) idx.push_back(i);
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return strcmp(classes[a].desc_name, classes[b].desc_name) < 0; });
    f << det << " " << idx.size() << std::endl;
    for (int ci : idx) {
      const modsx_region_class &c = classes[ci];
      f << c.desc_name << " " << c.n << std::endl;
      if (c.n > 0) f << (size_t)c.dim << std::endl;
      for (int i = 0; i < c.n; i++) {
        const modsx_region &r = c.regs[i];
        f << r.id << " " << r.img_id << " " << r.img_reproj_id << " ";     // saveAR, :89-99
        f << r.parent_id << " ";
        save_kp(r.det_kp, f);
        save_kp(r.reproj_kp, f);
        f << " " << (size_t)c.dim << " ";
        const float *d = c.desc + (size_t)i * c.stride;
        for (int q = 0; q < c.dim; q++) f << d[q] << " ";
        f << std::endl;
      }
    }
  }
  f.close();
  return f.fail() ? MODSX_ERR_ARG : MODSX_OK;
}

// Reads the class (det_name, desc_name); empty names select the first class of the file.
int load_regions(const char *path, const char *det_name, const char *desc_name, std::vector<modsx_region> &regs,
                 std::vector<float> &desc, int *dim, std::string *found_det, std::string *found_desc) {
  regs.clear(); desc.clear(); *dim = 0;
  std::ifstream f(path);
  if (!f.is_open()) { set_error(std::string("cannot open ") + path + " to load keypoints"); return MODSX_ERR_ARG; }
  // counts come from the file: bound them before they size anything (a corrupt header must not reach reserve())
  const int MAX_CLASSES = 1 << 12, MAX_KP = 1 << 26, MAX_DIM = 1 << 16;
  auto bad = [&](const char *what) { set_error(std::string("malformed key file ") + path + ": " + what); return MODSX_ERR_ARG; };
  int ndet = 0;
  f >> ndet;
  if (!f || ndet < 0 || ndet > MAX_CLASSES) return bad("detector count");
  bool taken = false;
  for (int d = 0; d < ndet; d++) {
    std::string det;
    int ndesc = 0;
    f >> det >> ndesc;
    if (!f || ndesc < 0 || ndesc > MAX_CLASSES) return bad("descriptor class count");
    for (int q = 0; q < ndesc; q++) {
      std::string dn;
      int nkp = 0, dsize = 0;
      f >> dn >> nkp;
      if (!f || nkp < 0 || nkp > MAX_KP) return bad("region count");
      // SaveRegions omits the length line for an empty class (:2157-2161) while LoadRegions reads it unconditionally
      // (:2200-2201) and so mis-parses what follows an empty class; here the file is read the way it is written
      if (nkp > 0) f >> dsize;
      if (!f || dsize < 0 || dsize > MAX_DIM) return bad("descriptor length");
      const bool want = !taken && (!det_name || !*det_name || det == det_name) && (!desc_name || !*desc_name || dn == desc_name);
      if (want) { taken = true; *dim = dsize; *found_det = det; *found_desc = dn; regs.reserve(nkp); desc.reserve((size_t)nkp * dsize); }
      for (int i = 0; i < nkp; i++) {
        modsx_region r;
        memset(&r, 0, sizeof r);
        f >> r.id >> r.img_id >> r.img_reproj_id;      // loadAR, :134-147
        f >> r.parent_id;
        load_kp(r.det_kp, f);
        load_kp(r.reproj_kp, f);
        int sz = 0;
        f >> sz;
        if (!f || sz < 0 || sz > MAX_DIM) return bad("truncated or corrupt region record");
        for (int k = 0; k < sz; k++) {
          float v;
          f >> v;
          if (want && k < dsize) desc.push_back(v);
        }
        if (!f) return bad("truncated descriptor");   // EOF inside a record is an error, not a short file
        if (want) { for (int k = sz; k < dsize; k++) desc.push_back(0.f); regs.push_back(r); }
      }
    }
  }
  if (!taken) { set_error("requested detector/descriptor class not in the key file"); return MODSX_ERR_ARG; }
  return MODSX_OK;
}

}  // namespace mx
