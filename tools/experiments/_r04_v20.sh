bash tools/_r04_v19.sh
bash tools/_r04_v18.sh
