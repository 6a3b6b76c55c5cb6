// C-ABI shim over the REFERENCE's own ConfigParser (include/parser/parser.h:39-50,
// src/parser.cpp), compiled from /root/reference by oracle/Makefile into
// oracle/_ref/.  Test infrastructure: lets tests compare the product's
// settings.config parser field by field with the real one.  Only this shim is
// ours; no reference source is copied.
#include "parser/parser.h"

#include <cstdint>
#include <cstring>

extern "C" {

struct ref_config_c {
  int32_t mode;
  uint64_t ncols, nrows, nb_of_obstacles, minWidth, maxWidth, minHeight, maxHeight;
  int32_t randomSeed, seedValue;
  char imagePath[1024];
  int32_t start_x, start_y, end_x, end_y;
  uint64_t max_iter;
  double visibilityThreshold;
  float lightStrength;
  int32_t timer, saveResults, saveLocalVisibility, saveCameFrom, saveLightSources,
      saveGlobalVisibility, saveVisibilityField, silent, ballRadius;
};

// returns 1 when parse() returned true, 0 otherwise
int ref_parse_config(const char* path, ref_config_c* out) {
  vbs::ConfigParser p;
  const bool ok = p.parse(path);
  const vbs::Config& c = p.getConfig();
  std::memset(out, 0, sizeof(*out));
  out->mode = c.mode;
  out->ncols = c.ncols; out->nrows = c.nrows; out->nb_of_obstacles = c.nb_of_obstacles;
  out->minWidth = c.minWidth; out->maxWidth = c.maxWidth;
  out->minHeight = c.minHeight; out->maxHeight = c.maxHeight;
  out->randomSeed = c.randomSeed; out->seedValue = c.seedValue;
  std::strncpy(out->imagePath, c.imagePath.c_str(), sizeof(out->imagePath) - 1);
  out->start_x = c.start.first; out->start_y = c.start.second;
  out->end_x = c.end.first; out->end_y = c.end.second;
  out->max_iter = c.max_iter;
  out->visibilityThreshold = c.visibilityThreshold;
  out->lightStrength = c.lightStrength;
  out->timer = c.timer; out->saveResults = c.saveResults;
  out->saveLocalVisibility = c.saveLocalVisibility; out->saveCameFrom = c.saveCameFrom;
  out->saveLightSources = c.saveLightSources; out->saveGlobalVisibility = c.saveGlobalVisibility;
  out->saveVisibilityField = c.saveVisibilityField; out->silent = c.silent;
  out->ballRadius = c.ballRadius;
  return ok ? 1 : 0;
}

}
