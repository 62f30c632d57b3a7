# parity tests (pytest -k "$AB_TESTS" over tests/, default: none), then library-variant A/B on ONE box, two interleaved rounds:
#   tools/ab_env.sh MATCH TAG ...     ("" = product library; NAME=VALUE = the product library under that environment variable; MATCH = comma list of kernel-label substrings to print)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
match=$1; shift
if [ -n "$AB_TESTS" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$AB_TESTS" > gpurun_out/ab/test.log 2>&1 || { tail -40 gpurun_out/ab/test.log; exit 1; }
  tail -2 gpurun_out/ab/test.log
fi
names=""
for r in 1 2; do
  for tag in "$@"; do
    case "$tag" in
      *=*)   # NAME=VALUE: the product library under that environment variable
        name=$(echo "$tag" | tr -c 'A-Za-z0-9\n' '_')
        env "$tag" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/ab/${name}_$r.json 2> gpurun_out/ab/${name}_$r.err || exit 1
        names="$names ${name}_$r" ;;
      *)
        lib=$PWD/video-frame-interpolation_amd/emavfi/lib/libemavfi${tag:+_$tag}.so
        EMAVFI_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/ab/${tag:-prod}_$r.json 2> gpurun_out/ab/${tag:-prod}_$r.err || exit 1
        names="$names ${tag:-prod}_$r" ;;
    esac
  done
done
python tools/ab_print.py gpurun_out/ab $names --match=$match
