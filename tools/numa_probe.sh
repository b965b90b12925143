for f in /sys/class/drm/card*/device/numa_node; do echo $f $(cat $f) $(cat $(dirname $f)/vendor); done 2>/dev/null | head
ls /sys/devices/system/node/ | grep node | head
for n in /sys/devices/system/node/node*; do echo $n $(cat $n/cpulist); done
python3 - <<'PY'
import torch
p=torch.cuda.get_device_properties(0)
print([a for a in dir(p) if 'pci' in a], getattr(p,'pci_bus_id',None), getattr(p,'pci_device_id',None), getattr(p,'pci_domain_id',None))
PY
