# eager cfg2 step, host side: raw stream query vs torch.cuda.current_stream() (same box, alternating)
for rep in 1 2 3; do
  GKG_DISABLE=raw_stream python tools/debug/eager_profile.py 2>&1 | grep "eager ms/step" | sed 's/^/current_stream() /'
  python tools/debug/eager_profile.py 2>&1 | grep "eager ms/step" | sed 's/^/raw query        /'
done
