// Test double of the reference's Node API TCP listeners (api/server.js:123-136, same rule for
// the map and detection ports): chunks are appended until the buffer ends with "}", then the
// buffer IS the document the HTTP routes serve (/api/map, /api/detection).  The real server.js
// needs express (not installed here); this file restates only its framing rule and then parses
// each document the way html/js/plot_map.js consumes it.
//   node frame_consumer.js <nFrames>     -> prints "PORT <p>" then one summary line per frame
const net = require('net');
const nFrames = parseInt(process.argv[2] || '1');
let data_map = '';
let seen = 0;
const server = net.createServer((socket) => {
  socket.on('data', (msg) => {
    data_map = data_map + msg.toString();
    if (data_map.slice(-1) === '}') {
      const map = data_map;
      data_map = '';
      const doc = JSON.parse(map);
      const s = { bytes: map.length, keys: Object.keys(doc), timestamp: doc.timestamp };
      if (doc.data) {
        s.nRows = doc.nRows; s.nCols = doc.nCols; s.rows = doc.data.length; s.cols = doc.data[0].length;
        s.noisePower = doc.noisePower; s.maxPower = doc.maxPower;
        s.delay0 = doc.delay[0]; s.dopplerLast = doc.doppler[doc.doppler.length - 1];
        let mx = -1e300; for (const r of doc.data) for (const v of r) if (v > mx) mx = v;
        s.dataMax = mx;
      } else {
        s.nDetections = doc.delay.length; s.delay = doc.delay; s.doppler = doc.doppler; s.snr = doc.snr;
      }
      console.log(JSON.stringify(s));
      if (++seen >= nFrames) { server.close(); socket.destroy(); }
    }
  });
});
server.listen(0, '127.0.0.1', () => console.log('PORT ' + server.address().port));
