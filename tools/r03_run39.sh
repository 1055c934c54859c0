b() { echo "$1 bench: $(timeout -k 10 200 python bench.py --no-cpu-baseline --set $2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
python - <<'P'
import sys; sys.path.insert(0,'.')
from fypraytracer_amd import capi, scenes
ctx=capi.Context(0); ctx.resize(1920,1080); ctx.upload_scene(scenes.hall_scene()); print("auto budget", ctx.get_tuning(8)); ctx.close()
P
for rep in 1 2; do
b auto 8=0
b b16 8=16
b b18 8=18
b b20 8=20
b b21 8=21
b b24 8=24
b b28 8=28
done
