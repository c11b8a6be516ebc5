bash tools/_r04_v12.sh
bash tools/_r04_v10.sh
