/*
 * common/log.h -- host-side call logging controlled by the environment, as the
 * reference documents it (README.md:79-88, CHANGELOG.md:19):
 *   NVCOMP_LOG_LEVEL  0 (off, default) .. 5; from level 3 up every low-level call is logged
 *   NVCOMP_LOG_FILE   a path, "stdout" or "stderr"; default nvcomp_yyyy-mm-dd_hh-mm.log
 */
#pragma once

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

namespace nvlog {

inline int level()
{
  static const int lv = [] {
    const char* e = getenv("NVCOMP_LOG_LEVEL");
    return e ? atoi(e) : 0;
  }();
  return lv;
}

inline FILE* sink()
{
  static FILE* f = [] {
    const char* e = getenv("NVCOMP_LOG_FILE");
    if (e != nullptr && strcmp(e, "stdout") == 0) {
      return stdout;
    }
    if (e != nullptr && strcmp(e, "stderr") == 0) {
      return stderr;
    }
    char name[64];
    if (e == nullptr) {
      const time_t t = time(nullptr);
      struct tm tmv;
      localtime_r(&t, &tmv);
      strftime(name, sizeof(name), "nvcomp_%Y-%m-%d_%H-%M.log", &tmv);
      e = name;
    }
    FILE* out = fopen(e, "a");
    return out ? out : stderr;
  }();
  return f;
}

/* log one API call at `lv` (3 = every low-level call, 1 = errors) */
inline void call(int lv, const char* fmt, ...)
{
  if (level() < lv) {
    return;
  }
  FILE* f = sink();
  va_list ap;
  va_start(ap, fmt);
  fputs("[nvcomp] ", f);
  vfprintf(f, fmt, ap);
  fputc('\n', f);
  fflush(f);
  va_end(ap);
}

} // namespace nvlog
