# Kernel times of alternative builds of the library, without the parity tests (see variant_check.sh).
#   LIBS="variants/a.so variants/b.so" SCENES="field fieldxl" bash profiles/sweep_only.sh
set -e
for scene in ${SCENES:-field fieldxl}; do
  bash $GRAFT_REPO_ROOT/profiles/sweep_lib.sh "$LIBS" $scene 24
done
