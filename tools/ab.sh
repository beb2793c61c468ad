#!/bin/bash
# A/B harness for kernel experiments.  HERE (no GPU):  tools/ab.sh build NAME "-DFLAG ..."   builds
# cosinesampler_amd/lib/alt_NAME.so from the working tree with extra flags (the .so travels with gpurun).
# ON THE GPU BOX:  tools/ab.sh run NAME [NAME...]   prints bench stage times for the default library and each variant.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  # UNIT: the unit(s) recompiled with the extra flags (default cs_abi; several separated by blanks), linked with the other
  # units' objects of the default build
  python -m cosinesampler_amd.build > /dev/null
  OBJS=""
  for u in cs_abi cs_coherent cs_coherent_sum cs_sort; do
    case " ${UNIT:-cs_abi} " in
      *" $u "*)
        # the unit's own flags of the default build (cosinesampler_amd/build.py EXTRA_FLAGS): a variant must differ from the
        # default library by its -D flags ONLY (round 4: variants of cs_abi were built with -fno-slp-vectorize, which the
        # default cs_abi is not, and lost 2.4 % of the drawn-points step to that alone -- profiles/round4_ablation.txt section 12)
        UFLAGS=$(python -c "from cosinesampler_amd import build; print(' '.join(build.COMMON_FLAGS + build.EXTRA_FLAGS.get('$u.hip', [])))" 2>/dev/null)
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wall -Wno-unused-function ${BASEFLAGS-$UFLAGS} $3 \
            -o $R/cosinesampler_amd/lib/obj/alt_$2_$u.o $R/cosinesampler_amd/csrc/$u.hip
        OBJS="$OBJS $R/cosinesampler_amd/lib/obj/alt_$2_$u.o" ;;
      *) OBJS="$OBJS $R/cosinesampler_amd/lib/obj/$u.o" ;;
    esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$R/cosinesampler_amd/csrc/exports.map \
      -o $R/cosinesampler_amd/lib/alt_$2.so $OBJS
  echo built alt_$2.so
elif [ "$1" = run ]; then
  shift
  for v in default "$@"; do
    if [ $v = default ]; then unset COSINESAMPLER_LIB; else export COSINESAMPLER_LIB=$R/cosinesampler_amd/lib/alt_$v.so; fi
    python $R/bench.py --no-cpu-baseline --no-helmholtz --steps ${STEPS:-20} --warmup 3 > /tmp/ab_$v.json
    python - $v /tmp/ab_$v.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-16s step %.3f ms | " % (sys.argv[1], j["ms_per_step"]) + "  ".join("%s %.3f" % (k[:9], v) for k, v in j["stages_ms"].items()), flush=True)
p = j.get("presorted_points")
if p:
    print("%-16s   ordered points %.3f ms | " % ("", p["ms_per_step"]) + "  ".join("%s %.3f" % (k[:9], v) for k, v in p["stages_ms"].items()), flush=True)
PY
  done
fi
