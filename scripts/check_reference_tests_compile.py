"""Type-check the reference's own gtest sources against the C++ host mirror (authoring container only: needs /root/reference).

Nothing of the reference is copied into the repo: each test/cpp/*.cpp is read where it lies, its `#include "<quake header>.h"`
lines are pointed at quake_amd/cpp/quake.h in a scratch copy under /tmp, FAISS includes are dropped (FAISS is not in the
image), a scratch gtest macro stub stands in for gtest (also not in the image) and `g++ -std=c++17 -fsyntax-only` runs.
What this shows: the class / method / member names and signatures the reference's tests use exist in the mirror.  It does
not run the tests (those run as tests/test_bindings_gpu.py on the GPU).  Result of the last run: see INTEGRATION.md section 3."""
import os
import re
import subprocess
import sys
import tempfile

REF = "/root/reference/test/cpp"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = ["quake_index", "query_coordinator", "list_scanning", "topk_buffer", "partition_manager", "maintenance", "hit_count_tracker",
         "latency_estimator", "maintenance_cost_estimator", "search_recall_tests"]
GTEST = r'''#pragma once
#include <cmath>
#include <string>
namespace testing { class Test { public: virtual ~Test() {} virtual void SetUp() {} virtual void TearDown() {} };
struct Msg { template <class T> Msg &operator<<(const T &) { return *this; } }; inline void InitGoogleTest(int *, char **) {} }
#define GT_CAT_(a, b) a##_##b
#define TEST(a, b) struct GT_CAT_(a, b) { void body(); }; void GT_CAT_(a, b)::body()
#define TEST_F(a, b) struct GT_CAT_(a, b) : public a { void body(); }; void GT_CAT_(a, b)::body()
#define GT_CHK_(x) if (x) {} else ::testing::Msg()
#define GT_BIN_(n, op) n(a, b) GT_CHK_((a)op(b))
'''
for n, op in (("EQ", "=="), ("NE", "!="), ("GT", ">"), ("GE", ">="), ("LT", "<"), ("LE", "<=")):
    GTEST += f"#define EXPECT_{n}(a, b) GT_CHK_((a){op}(b))\n#define ASSERT_{n}(a, b) GT_CHK_((a){op}(b))\n"
GTEST += r'''#define EXPECT_TRUE(a) GT_CHK_(!!(a))
#define ASSERT_TRUE(a) GT_CHK_(!!(a))
#define EXPECT_FALSE(a) GT_CHK_(!(a))
#define ASSERT_FALSE(a) GT_CHK_(!(a))
#define EXPECT_NEAR(a, b, t) GT_CHK_(std::fabs((double)(a) - (double)(b)) <= (t))
#define ASSERT_NEAR(a, b, t) GT_CHK_(std::fabs((double)(a) - (double)(b)) <= (t))
#define EXPECT_FLOAT_EQ(a, b) GT_CHK_((a) == (b))
#define ASSERT_FLOAT_EQ(a, b) GT_CHK_((a) == (b))
#define EXPECT_DOUBLE_EQ(a, b) GT_CHK_((a) == (b))
#define EXPECT_THROW(s, e) try { s; } catch (const e &) {} catch (...) {}
#define ASSERT_THROW(s, e) try { s; } catch (const e &) {} catch (...) {}
#define EXPECT_ANY_THROW(s) try { s; } catch (...) {}
#define EXPECT_NO_THROW(s) try { s; } catch (...) {}
#define ASSERT_NO_THROW(s) try { s; } catch (...) {}
#define SUCCEED() ::testing::Msg()
#define FAIL() ::testing::Msg()
#define GTEST_SKIP() ::testing::Msg()
#define RUN_ALL_TESTS() 0
'''

if not os.path.isdir(REF):
    sys.exit("reference checkout not present (this check only runs in the authoring container)")
from torch.utils import cpp_extension

inc = [f"-I{p}" for p in cpp_extension.include_paths()]
with tempfile.TemporaryDirectory() as tmp:
    os.makedirs(os.path.join(tmp, "gtest"))
    open(os.path.join(tmp, "gtest", "gtest.h"), "w").write(GTEST)
    ours = r"(quake_index|query_coordinator|partition_manager|list_scanning|common|maintenance_policies|maintenance_cost_estimator|hit_count_tracker|clustering|dynamic_inverted_list|index_partition)"
    for t in TESTS:
        src = open(os.path.join(REF, t + ".cpp")).read()
        src = re.sub(r'#include ["<]' + ours + r'\.h[">]', '#include "quake.h"', src)
        src = re.sub(r'#include ["<]faiss/[^">]*[">]', "// (faiss include dropped)", src)
        path = os.path.join(tmp, t + ".cpp")
        open(path, "w").write(src)
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", f"-I{tmp}", f"-I{ROOT}/quake_amd/cpp"] + inc + [path],
                           capture_output=True, text=True)
        errs = [ln for ln in r.stderr.splitlines() if " error: " in ln]
        print(f"{t}.cpp: {'OK' if not errs else str(len(errs)) + ' errors'}")
        for e in errs[:6]:
            print("    " + e.split(" error: ", 1)[1][:140])
