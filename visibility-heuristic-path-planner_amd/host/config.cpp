#include "config.hpp"

#include <cstdio>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>

namespace vbs {
namespace {

const char* const kBlank = " \t";

std::string trimmed(std::string s) {
  const auto a = s.find_first_not_of(kBlank);
  if (a == std::string::npos) return std::string();
  const auto b = s.find_last_not_of(kBlank);
  return s.substr(a, b - a + 1);
}

void complain(const std::string& key, const std::string& value) {
  std::cerr << "Invalid value for " << key << ": " << value << '\n';
}

// How one key is digested.  Returns false when the whole parse must fail.
using Handler = std::function<bool(Config&, const std::string& key, const std::string& value)>;

// non-negative integer into a size_t field; `lenient`: an unparsable value only warns
// (the reference forgets the `return false` for minHeight and max_iter, parser.cpp:125-127,183-185)
Handler count_field(std::size_t Config::*field, bool lenient) {
  return [=](Config& c, const std::string& k, const std::string& v) {
    try {
      const int n = std::stoi(v);
      if (n < 0) {
        complain(k, v);
        std::cerr << "It must be a positive integer\n";
        return false;
      }
      c.*field = n;
    } catch (...) {
      complain(k, v);
      if (lenient) return true;
      std::cerr << "It must be a positive integer\n";
      return false;
    }
    return true;
  };
}

Handler int_field(int Config::*field) {
  return [=](Config& c, const std::string& k, const std::string& v) {
    try {
      c.*field = std::stoi(v);
    } catch (...) {
      complain(k, v);
      std::cerr << "It must be an integer\n";
      return false;
    }
    return true;
  };
}

Handler flag_field(bool Config::*field) {
  return [=](Config& c, const std::string& k, const std::string& v) {
    if (v == "0" || v == "false") {
      c.*field = false;
    } else if (v == "1" || v == "true") {
      c.*field = true;
    } else {
      complain(k, v);
      std::cerr << "It must be a boolean\n";
      return false;
    }
    return true;
  };
}

// "{a,b}"; anything else warns and yields {0,0} (parser.cpp:343-353)
Handler pair_field(point Config::*field) {
  return [=](Config& c, const std::string&, const std::string& v) {
    point p;
    if (std::sscanf(v.c_str(), "{%d,%d}", &p.first, &p.second) != 2) {
      std::cerr << "Error: Invalid pair string: " << v << std::endl;
      p = {0, 0};
    }
    c.*field = p;
    return true;
  };
}

template <class T>
Handler unit_interval_field(T Config::*field) {
  return [=](Config& c, const std::string& k, const std::string& v) {
    try {
      c.*field = static_cast<T>(std::stod(v));
      if (c.*field > 1.0 || c.*field < 0.0) {
        complain(k, v);
        std::cerr << "It must be a double between 0 and 1\n";
        return false;
      }
    } catch (...) {
      complain(k, v);
      std::cerr << "It must be a positive double between 0 and 1\n";
      return false;
    }
    return true;
  };
}

const std::map<std::string, Handler>& handlers() {
  static const std::map<std::string, Handler> table = {
      {"mode",
       [](Config& c, const std::string& k, const std::string& v) {
         try {
           c.mode = std::stoi(v);
           if (c.mode != 1 && c.mode != 2) {
             std::cerr << "Invalid value for " << k << ": " << v << ", using default value 1\n";
             c.mode = 1;
           }
         } catch (...) {
           complain(k, v);
           std::cerr << "It must be an integer 1 or 2 \n";
           return false;
         }
         return true;
       }},
      {"ncols", count_field(&Config::ncols, false)},
      {"nrows", count_field(&Config::nrows, false)},
      {"nb_of_obstacles",
       [](Config& c, const std::string& k, const std::string& v) {
         try {
           c.nb_of_obstacles = std::stoi(v);  // negative values wrap, as in the reference (parser.cpp:90)
         } catch (...) {
           complain(k, v);
           std::cerr << "It must be an integer\n";
           return false;
         }
         return true;
       }},
      {"minWidth", count_field(&Config::minWidth, false)},
      {"maxWidth", count_field(&Config::maxWidth, false)},
      {"minHeight", count_field(&Config::minHeight, true)},
      {"maxHeight", count_field(&Config::maxHeight, false)},
      {"randomSeed", flag_field(&Config::randomSeed)},
      {"seedValue", int_field(&Config::seedValue)},
      {"imagePath",
       [](Config& c, const std::string&, const std::string& v) {
         c.imagePath = v;
         return true;
       }},
      {"start", pair_field(&Config::start)},
      {"end", pair_field(&Config::end)},
      {"max_iter", count_field(&Config::max_iter, true)},
      {"visibilityThreshold", unit_interval_field(&Config::visibilityThreshold)},
      {"lightStrength", unit_interval_field(&Config::lightStrength)},
      {"timer", flag_field(&Config::timer)},
      {"saveResults", flag_field(&Config::saveResults)},
      {"saveLocalVisibility", flag_field(&Config::saveLocalVisibility)},
      {"saveCameFrom", flag_field(&Config::saveCameFrom)},
      {"saveLightSources", flag_field(&Config::saveLightSources)},
      {"saveGlobalVisibility", flag_field(&Config::saveGlobalVisibility)},
      {"saveVisibilityField", flag_field(&Config::saveVisibilityField)},
      {"silent", flag_field(&Config::silent)},
      {"ballRadius", int_field(&Config::ballRadius)},
  };
  return table;
}

void echo(const Config& c) {
  std::ostream& o = std::cout;
  if (c.mode == 1) {
    o << "Random environment mode" << std::endl;
    o << "################### Environment settings ################## \n"
      << "nrows: " << c.nrows << "\n"
      << "ncols: " << c.ncols << "\n"
      << "Nb of obstacles: " << c.nb_of_obstacles << "\n"
      << "Min width: " << c.minWidth << "\n"
      << "Max width: " << c.maxWidth << "\n"
      << "Min height: " << c.minHeight << "\n"
      << "Max height: " << c.maxHeight << std::endl;
    if (c.randomSeed)
      o << "Random seed: " << c.randomSeed << std::endl;
    else
      o << "Fixed seed value: " << c.seedValue << std::endl;
  } else if (c.mode == 2) {
    o << "Import image mode\nImage path: " << c.imagePath << std::endl;
  }
  o << "#################### Solver settings ###################### \n"
    << "Start point: " << c.start.first << ", " << c.start.second << "\n"
    << "End point: " << c.end.first << ", " << c.end.second << "\n"
    << "Maximum iterations: " << c.max_iter << "\n"
    << "Solver visibility threshold: " << c.visibilityThreshold << "\n"
    << "Light strength: " << c.lightStrength << std::endl;
  // the labels below are the reference's (parser.cpp:327-335), including its
  // cross-wiring of saveGlobalVisibility / saveVisibilityField
  o << "#################### Output settings ###################### \n"
    << "timer: " << c.timer << "\n"
    << "saveLightSourceEnum: " << c.saveCameFrom << "\n"
    << "saveLightSources: " << c.saveLightSources << "\n"
    << "saveVisibilityField: " << c.saveGlobalVisibility << "\n"
    << "saveLocalVisibility: " << c.saveLocalVisibility << "\n"
    << "saveVisibilityMapEnv: " << c.saveVisibilityField << std::endl;
}

}  // namespace

bool ConfigParser::parse(const std::string& filename) {
  std::ifstream in(filename);
  if (!in) {
    std::cerr << "Failed to open " << filename << '\n';
    return false;
  }
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty() || line.front() == '#') continue;  // only a '#' in column 0 starts a comment
    const auto eq = line.find('=');
    // no '=' at all, or nothing after it: the reference's second getline fails and the line is dropped
    if (eq == std::string::npos || eq + 1 >= line.size()) continue;
    const std::string key = trimmed(line.substr(0, eq));
    const std::string value = trimmed(line.substr(eq + 1));
    const auto it = handlers().find(key);
    if (it == handlers().end()) {
      std::cerr << "Invalid/irrelavent key: " << key << '\n';
      continue;
    }
    if (!it->second(config_, key, value)) return false;
  }
  if (!config_.silent) echo(config_);
  return true;
}

}  // namespace vbs
