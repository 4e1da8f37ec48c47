"""Host CLI number parsing: fast_strtod() (ngsdist_host.cpp) must return exactly what strtod returns, since the
reference's split() (gen_func.cpp:390-417) converts every field with strtod.  The check compiles the host source
with a test main (no GPU, no engine calls)."""
import os
import subprocess
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ngsdist_amd", "csrc", "host", "ngsdist_host.cpp")
LIBDIR = os.path.join(ROOT, "ngsdist_amd")


def test_fast_strtod_equals_strtod(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(textwrap.dedent(r'''
        #include <random>
        #define main host_main
        #include "%s"
        #undef main
        int main() {
          std::mt19937_64 r(1); char b[128]; unsigned long bad = 0, fast = 0, n = 0;
          auto chk = [&](const char *s) {
            double v; n++;
            if (fast_strtod(s, strlen(s), &v)) {
              fast++; char *e; double w = strtod(s, &e);
              if (*e || memcmp(&v, &w, 8)) { if (bad < 10) printf("MISMATCH %%s %%a %%a\n", s, v, w); bad++; }
            }
          };
          const char *fixed[] = {"0", "-0", "0.0", "-0.0", "1", "2", "-1", "0.333333", "1e-5", "1E5", "1e22", "1e23",
            "123456789012345678", "1234567890123456789", "12345678901234567890", ".5", "5.", "-.5e-3", "+3.25",
            "1e", "e5", ".", "", "-", "nan", "inf", "0x10", "1e-22", "1e-23", "9007199254740992", "9007199254740993",
            "0.1e1", "000.000100", "1.7976931348623157e308", "4.9e-324", "0e999999", "1e+", "--1", "1.2.3"};
          for (auto s : fixed) chk(s);
          for (int i = 0; i < 400000; i++) {
            int k = r() %% 6; double x;
            switch (k) {
              case 0: x = (double)(r() %% 1000000) / 1e6; snprintf(b, 128, "%%.6f", x); break;
              case 1: x = std::ldexp((double)(r() >> 11), -53); snprintf(b, 128, "%%.17g", x); break;
              case 2: x = std::ldexp((double)(r() >> 11), -53); snprintf(b, 128, "%%.10g", x); break;
              case 3: x = std::ldexp((double)(r() >> 11), -53) * pow(10, (int)(r() %% 60) - 30);
                      snprintf(b, 128, "%%.15e", x); break;
              case 4: x = std::ldexp((double)(r() >> 11), -53); snprintf(b, 128, "%%.12f", -x); break;
              default: snprintf(b, 128, "%%llu.%%llue%%d", (unsigned long long)(r() %% 100000),
                                (unsigned long long)(r() %% 1000000000), (int)(r() %% 50) - 25);
            }
            chk(b);
          }
          printf("n=%%lu fast=%%lu bad=%%lu\n", n, fast, bad);
          return bad != 0 || fast < n / 2;
        }
        ''' % HOST))
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O1", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L" + LIBDIR, "-lngsdist_amd", "-lz", "-lpthread", "-Wl,-rpath," + LIBDIR], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
