#!/bin/bash
# What can an ordinary user on the GPU box read about the GPU's clock, power and temperature, and how long does one sample take?
# (the sustained leg of bench.py samples them every 100 ms: VERDICT r5 item 1)
out=gpurun_out/clock_probe.txt
{
echo "== ls /sys/class/drm"; ls /sys/class/drm 2>&1
for c in /sys/class/drm/card*/device; do
  [ -d "$c" ] || continue
  echo "== $c"; cat $c/vendor $c/device 2>&1 | tr '\n' ' '; echo
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk gpu_busy_percent mem_busy_percent power_dpm_force_performance_level current_link_speed unique_id; do
    echo "-- $f"; cat $c/$f 2>&1 | head -20
  done
  for h in $c/hwmon/hwmon*; do
    echo "-- $h"; ls $h 2>&1 | tr '\n' ' '; echo
    for f in name freq1_input freq1_label freq2_input freq2_label power1_average power1_input power1_cap temp1_input temp1_label temp2_input temp2_label temp3_input temp3_label in0_input; do
      [ -e $h/$f ] && { echo -n "$f: "; cat $h/$f 2>&1; }
    done
  done
done
echo "== rocm-smi"; which rocm-smi amd-smi 2>&1
( time rocm-smi --showclocks --showpower --showtemp --showuse --json ) 2>&1 | head -60
echo "== amd-smi"; ( time amd-smi metric --json ) 2>&1 | head -120
echo "== python amdsmi"; python3 -c "import amdsmi; print(amdsmi.__file__)" 2>&1 | tail -1
python3 - <<'PY' 2>&1
import sys
for p in ("/opt/rocm/share/amd_smi", "/opt/rocm/libexec/rocm_smi"):
    sys.path.insert(0, p)
try:
    import amdsmi, time
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    print("handles", len(hs))
    t = time.perf_counter()
    for _ in range(10):
        m = amdsmi.amdsmi_get_gpu_metrics_info(hs[0])
    print("gpu_metrics_info: %.2f ms per call" % ((time.perf_counter() - t) * 100))
    print({k: m[k] for k in m if any(s in k for s in ("gfxclk", "power", "temperature_hotspot", "temperature_edge", "average_gfx_activity", "current_socket"))})
    try:
        print("clock", amdsmi.amdsmi_get_clock_info(hs[0], amdsmi.AmdSmiClkType.GFX))
    except Exception as e:
        print("clock_info:", type(e).__name__, e)
    try:
        print("power", amdsmi.amdsmi_get_power_info(hs[0]))
    except Exception as e:
        print("power_info:", type(e).__name__, e)
except Exception as e:
    print("amdsmi python:", type(e).__name__, e)
PY
echo "== librocm_smi64"; ls /opt/rocm/lib/librocm_smi64.so* /opt/rocm/lib/libamd_smi.so* 2>&1
} > $out 2>&1
echo done
